"""ctypes binding of libmugd.so (the C ABI in include/mugd.h).

PyTorch is used only as the owner of device memory: every call passes raw
``tensor.data_ptr()`` values.  The library works on a stream of its own (the legacy NULL stream
cannot be captured into a hipGraph) and every call is bracketed by event waits against
``torch.cuda.current_stream()`` (`mugd_order_after` / `mugd_order_before`), so torch ops before and
after a call are ordered with it on whatever stream the caller uses -- no host synchronisation.  There
is NO CPU fallback: if the in-tree HIP library is missing or no GPU is visible,
``get_lib()`` raises.

(``Lib(path=..., device='cpu')`` can also be pointed at another build of the same
C ABI; the test-suite uses that to drive tests/emu/libmugd_emu.so, the functional
CPU emulation of the kernels.  Product code never does.)
"""
import ctypes as C
import os

import numpy as np
import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MUGD_LIB_PATH") or os.path.join(os.path.dirname(_HERE), "libmugd.so")      # MUGD_LIB_PATH: development A/B builds (build.py --variant)

F32, I64 = 0, 1
_p = C.c_void_p
_i = C.c_int
_f = C.c_float


class UNetConfig(C.Structure):
    _fields_ = [("in_channels", _i), ("model_channels", _i), ("out_channels", _i), ("num_res_blocks", _i),
                ("n_levels", _i), ("channel_mult", _i * 8),
                ("n_attn", _i), ("attention_resolutions", _i * 8),
                ("num_heads", _i), ("context_dim", _i),
                ("audio_channels", _i * 8), ("s4_layer", _i)]


class VaeConfig(C.Structure):
    _fields_ = [("x_channels", _i), ("middle_channels", _i), ("z_channels", _i), ("num_groups", _i),
                ("num_res_blocks", _i), ("n_levels", _i), ("channel_mult", _i * 8), ("scale", _f)]


class WaveConfig(C.Structure):
    _fields_ = [("n_freq", _i), ("middle_channels", _i), ("num_res_blocks", _i), ("num_heads", _i),
                ("num_groups", _i), ("n_levels", _i), ("channel_mult", _i * 16),
                ("n_attn", _i), ("attention_resolutions", _i * 8)]


_SIGS = {
    "mugd_create": [_i, _p, C.POINTER(_p)],
    "mugd_synchronize": [_p],
    "mugd_order_after": [_p, _p],
    "mugd_order_before": [_p, _p],
    "mugd_set_graph_mode": [_p, _i],
    "mugd_set_conv_tiling": [_p, _i, _i],
    "mugd_set_s4_symmetric": [_p, _i],
    "mugd_set_mel_pad_mode": [_p, _i],
    "mugd_set_weight_precision": [_p, _i],
    "mugd_unet_create": [_p, C.POINTER(UNetConfig), C.POINTER(_p)],
    "mugd_vae_create": [_p, C.POINTER(VaeConfig), C.POINTER(_p)],
    "mugd_wave_create": [_p, C.POINTER(WaveConfig), C.POINTER(_p)],
    "mugd_vae_encoder_create": [_p, C.POINTER(VaeConfig), C.POINTER(_p)],
    "mugd_vae_encode": [_p, _p, _p, _i, _i],
    "mugd_net_set_param": [_p, C.c_char_p, _p, _i, _i, C.POINTER(C.c_int64)],
    "mugd_net_invalidate": [_p],
    "mugd_unet_forward": [_p, _p, _p, _p, _i, C.POINTER(_p), _i, _p, _i, _i],
    "mugd_ddim_sample": [_p, _p, _p, _p, _i, C.POINTER(_p), _i, _i, _i, _i, C.POINTER(C.c_int64), C.POINTER(_f), _f, _p, _p, _p],
    "mugd_net_profile": [_p, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_int64)],
    "mugd_net_host_enqueue": [_p, _i, C.POINTER(C.c_double), C.POINTER(C.c_int64)],
    "mugd_vae_decode": [_p, _p, _p, _i, _i],
    "mugd_wave_encode": [_p, _p, C.POINTER(_p), _i, _i],
    "mugd_cond_embed": [_p, _p, _p, _p, _i, _i, _i],
    "mugd_log_mel": [_p, _p, C.c_int64, _i, _i, _i, _i, _p],
    "mugd_resample_poly": [_p, _p, C.c_int64, _i, _i, _p, C.POINTER(C.c_int64)],
    "mugd_timing_sweep": [_p, _p, _i, _p, _p, _p, _i, C.c_double, _p],
    "mugd_remove_mini_jacks": [_i, _p, _p, _p, C.c_double, _i, _p, _p],
    "mugd_op_group_norm": [_p, _p, _p, _p, _p, _i, _i, _i, _i, _i],
    "mugd_op_layer_norm": [_p, _p, _p, _p, _p, _i, _i, _i],
    "mugd_op_conv1d": [_p, _p, _p, _p, _p, _p] + [_i] * 11,
    "mugd_op_norm_conv1d": [_p, _p, _p, _p, _p, _p, _p] + [_i] * 11,
    "mugd_dev_bench_conv": [_p] + [_i] * 11 + [C.POINTER(_f)],
    "mugd_dev_clock_probe": [_p, C.POINTER(_f)],
    "mugd_op_attention": [_p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i],
    "mugd_op_s4_kernel": [_p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i],
    "mugd_op_s4_conv": [_p, _p, _p, _p, _p, _i, _i, _i],
    "mugd_op_gn_s4_conv": [_p, _p, _p, _p, _p, _p, _i, _p, _i, _i, _i],
    "mugd_op_timestep_embedding": [_p, _p, _p, _i, _i],
    "mugd_train_q_sample": [_p, _p, _p, _p, _p, _p, _p, _i, C.c_int64],
    "mugd_train_smooth_l1": [_p, _p, _p, _f, _f, _p, _p, _i, C.c_int64],
    "mugd_train_resblock": [_p, _p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _p],
    "mugd_train_release_states": [_p],
    "mugd_train_set_precision": [_p, _i],
    "mugd_train_profile": [_p, _i, C.POINTER(C.c_double)],
    "mugd_train_step_begin": [_p],
    "mugd_train_step_flush": [_p],
    "mugd_train_step_end": [_p],
    "mugd_train_step_reset": [_p],
    "mugd_train_concat": [_p, _p, _p, _p, _i, _i, _i, _i],
    "mugd_train_split": [_p, _p, _p, _p, _i, _i, _i, _i, _i, _i],
    "mugd_train_add": [_p, _p, _p, _p, C.c_int64],
    "mugd_train_adamw_multi": [_p, _i, _p, _p, _p, _p, _p, _f, _f, _f, _f, _f, _i],
    "mugd_train_adamw_chunks": [_p, _p, _i, _f, _f, _f, _f, _f, _i],
    "mugd_train_conv": [_p] * 13 + [_i] * 8 + [_p],
    "mugd_train_resnet_block": [_p] * 7 + [_i] * 7 + [_p],
    "mugd_train_time_embed": [_p] * 12 + [_i] * 3,
    "mugd_train_embedding_bwd": [_p] * 4 + [_i] * 4,
    "mugd_train_s4layer": [_p] * 7 + [_i] * 6 + [_p],
    "mugd_train_transformer": [_p, _p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _i, _p],
    "mugd_train_adamw": [_p, _p, _p, _p, _p, C.c_int64, _f, _f, _f, _f, _f, _i],
}


class TrainState:
    """Handle of the forward intermediates a training block keeps between its forward-only call and its backward call
    (include/mugd.h: the `state` argument of the block entry points)."""

    def __init__(self):
        self.v = C.c_longlong(0)

    def ref(self):
        return C.cast(C.byref(self.v), C.c_void_p)


def _sref(state):
    return _p(None) if state is None else state.ref()


class ResBlockPtrs(C.Structure):            # mugd_resblock_params / mugd_resblock_grads (include/mugd.h): 12 pointers
    NAMES = ("gn1_w", "gn1_b", "conv1_w", "conv1_b", "emb_w", "emb_b", "gn2_w", "gn2_b", "conv2_w", "conv2_b", "skip_w", "skip_b")
    _fields_ = [(n, _p) for n in NAMES]
EXPORTS = sorted(list(_SIGS) + ["mugd_destroy", "mugd_net_destroy", "mugd_last_error", "mugd_version",
                                "mugd_profile_kind_name", "mugd_get_stream"])
PROFILE_KINDS = 7


def _ptr(t):
    return _p(t.data_ptr()) if t is not None else _p(None)


def _ilist(ctype_arr, values):
    for k, v in enumerate(values):
        ctype_arr[k] = int(v)


class MugdError(RuntimeError):
    pass


_UNORDERED = {"mugd_get_stream", "mugd_create", "mugd_destroy", "mugd_net_destroy", "mugd_last_error", "mugd_version", "mugd_profile_kind_name",
              "mugd_order_after", "mugd_order_before", "mugd_set_graph_mode", "mugd_set_conv_tiling", "mugd_set_s4_symmetric", "mugd_set_mel_pad_mode", "mugd_set_weight_precision", "mugd_remove_mini_jacks",
              "mugd_train_set_precision"}


class _OrderedDll:
    """The ctypes library with every enqueueing entry point bracketed by stream-ordering calls against torch's current
    stream (see the module docstring).  `.raw` is the plain CDLL."""

    def __init__(self, raw, lib):
        self.raw, self._lib, self._cache = raw, lib, {}

    def __getattr__(self, name):
        fn = getattr(self.raw, name)
        if name in _UNORDERED or not name.startswith("mugd_"):
            return fn
        if name not in self._cache:
            lib, raw = self._lib, self.raw

            def ordered(*args):
                if lib.device.type != "cuda" or not lib.ctx:
                    return fn(*args)
                cur = torch.cuda.current_stream(lib.device).cuda_stream
                if cur == lib.stream_ptr:                 # torch already runs on the library's stream (Lib.on_stream()): nothing to order
                    return fn(*args)
                cur = _p(cur)
                raw.mugd_order_after(lib.ctx, cur)
                rc = fn(*args)
                raw.mugd_order_before(lib.ctx, cur)
                return rc
            self._cache[name] = ordered
        return self._cache[name]


class Lib:
    def __init__(self, path=None, device=None):
        path = path or LIB_PATH
        if not os.path.exists(path):
            raise MugdError("%s not found: build it with `python mug-diffusion_amd/build.py` "
                            "(hipcc --offload-arch=gfx950); there is no CPU fallback" % path)
        self.path = path
        self.dll = _OrderedDll(C.CDLL(path), self)
        # development A/B runs load OLDER libraries through MUGD_LIB_PATH (tests/gpu_run.sh ab:...): entry points they lack stay unbound
        # there (calling one raises); the product library must export every declared symbol -- an ABI mismatch fails here, loudly
        lenient = bool(os.environ.get("MUGD_LIB_PATH")) and os.environ.get("MUGD_LIB_LENIENT", "") == "1"
        for name, args in _SIGS.items():
            try:
                fn = getattr(self.dll.raw, name)
            except AttributeError:
                if lenient:
                    continue
                raise
            fn.argtypes = args
            fn.restype = _i
        raw = self.dll.raw
        raw.mugd_last_error.argtypes = [_p]
        raw.mugd_last_error.restype = C.c_char_p
        raw.mugd_version.restype = C.c_char_p
        raw.mugd_profile_kind_name.argtypes = [_i]
        raw.mugd_profile_kind_name.restype = C.c_char_p
        raw.mugd_destroy.argtypes = [_p]
        raw.mugd_destroy.restype = None
        raw.mugd_net_destroy.argtypes = [_p]
        raw.mugd_net_destroy.restype = None
        if device is None:
            if not torch.cuda.is_available():
                raise MugdError("no GPU visible to torch: libmugd.so needs an MI355X (no CPU fallback)")
            device = torch.device("cuda", torch.cuda.current_device())
        self.device = torch.device(device)
        index = 0
        if self.device.type == "cuda":
            index = self.device.index or 0
        self.ctx = _p()
        rc = raw.mugd_create(index, _p(None), C.byref(self.ctx))          # NULL: the library creates its own stream
        if rc != 0:
            raise MugdError("mugd_create failed with status %d" % rc)
        self._nets = []
        raw.mugd_get_stream.argtypes = [_p]
        raw.mugd_get_stream.restype = _p
        self.stream_ptr = raw.mugd_get_stream(self.ctx) or 0
        self._ext_stream = None

    def on_stream(self):
        """Context manager: torch's current stream := the library's stream for the duration (tensors torch allocates / fills inside are
        then ordered with the library's kernels by stream order alone -- no event per call; a training step makes ~700 native calls).
        The library's stream first waits for what the caller's stream has queued, and the caller's stream waits for it at exit."""
        import contextlib
        if self.device.type != "cuda" or not self.stream_ptr:
            return contextlib.nullcontext()
        if self._ext_stream is None:
            self._ext_stream = torch.cuda.ExternalStream(self.stream_ptr, device=self.device)
        ext = self._ext_stream
        lib_dev = self.device

        @contextlib.contextmanager
        def cm():
            prev = torch.cuda.current_stream(lib_dev)
            if prev.cuda_stream == ext.cuda_stream:
                yield
                return
            ext.wait_stream(prev)
            try:
                with torch.cuda.stream(ext):
                    yield
            finally:                       # also when the body raises: the caller's stream stays ordered behind what the library queued
                prev.wait_stream(ext)
        return cm()

    # ------------------------------------------------------------------ plumbing
    def check(self, rc):
        if rc != 0:
            msg = self.dll.mugd_last_error(self.ctx)
            raise MugdError("mugd status %d: %s" % (rc, (msg or b"").decode(errors="replace")))

    def version(self):
        return self.dll.mugd_version().decode()

    def synchronize(self):
        self.check(self.dll.mugd_synchronize(self.ctx))

    def set_graph_mode(self, on):
        """True / 1: one hipGraph per DDIM step, replayed S times; False / 0: eager launches; 2: the whole loop as one graph."""
        self.check(self.dll.mugd_set_graph_mode(self.ctx, int(on)))

    def set_weight_precision(self, bf16):
        """Reduced-precision mode: networks compiled afterwards keep their packed conv / linear weights in bfloat16
        (include/mugd.h).  Call Net.set_params / invalidate to recompile existing ones."""
        self.check(self.dll.mugd_set_weight_precision(self.ctx, 1 if bf16 else 0))

    def set_s4_symmetric(self, on):
        """Cauchy sum over both conjugate halves of the S4 poles (reference: pykeops / CUDA-extension backends) instead of the
        stored half only (reference: cauchy_naive).  Affects S4 kernels baked afterwards."""
        self.check(self.dll.mugd_set_s4_symmetric(self.ctx, 1 if on else 0))

    def set_conv_tiling(self, wk=0, tn=0):
        """Force the conv_gemm K-split / tile width (0 = per-layer choice); affects networks compiled afterwards."""
        self.check(self.dll.mugd_set_conv_tiling(self.ctx, wk, tn))

    def f32(self, t):
        return t.detach().to(device=self.device, dtype=torch.float32).contiguous()

    # parameter-gradient tensors of the training entry points: zeroed, shaped like the parameter.  A training step installs a
    # FlatGrads arena (mug/train.py) so that a whole step's 1327 gradient tensors are views of ONE buffer cleared by ONE fill
    # instead of 1327 allocations + fills; without it every gradient is its own tensor.
    grad_arena = None

    def zgrad(self, like):
        ar = self.grad_arena
        if ar is not None:
            g = ar.take(like)
            if g is not None:
                return g
        return torch.zeros_like(like)

    def empty(self, *shape):
        return torch.empty(*shape, dtype=torch.float32, device=self.device)

    def close(self):
        if self.ctx:
            for n in self._nets:
                n.close()
            self.dll.mugd_destroy(self.ctx)
            self.ctx = _p()

    # ------------------------------------------------------------------ networks
    def unet(self, cfg):
        c = UNetConfig()
        c.in_channels, c.model_channels, c.out_channels = cfg["in_channels"], cfg["model_channels"], cfg["out_channels"]
        c.num_res_blocks = cfg["num_res_blocks"]
        c.n_levels = len(cfg["channel_mult"])
        _ilist(c.channel_mult, cfg["channel_mult"])
        c.n_attn = len(cfg["attention_resolutions"])
        _ilist(c.attention_resolutions, cfg["attention_resolutions"])
        c.num_heads, c.context_dim = cfg["num_heads"], cfg["context_dim"]
        _ilist(c.audio_channels, cfg["audio_channels"])
        c.s4_layer = 1 if cfg.get("s4_layer", False) else 0
        h = _p()
        self.check(self.dll.mugd_unet_create(self.ctx, C.byref(c), C.byref(h)))
        n = Net(self, h, "unet", dict(cfg))
        self._nets.append(n)
        return n

    def vae(self, cfg, scale=1.0, encoder=False):
        c = VaeConfig()
        c.x_channels, c.middle_channels, c.z_channels = cfg["x_channels"], cfg["middle_channels"], cfg["z_channels"]
        c.num_groups, c.num_res_blocks = cfg["num_groups"], cfg["num_res_blocks"]
        c.n_levels = len(cfg["channel_mult"])
        _ilist(c.channel_mult, cfg["channel_mult"])
        c.scale = float(scale)
        h = _p()
        self.check((self.dll.mugd_vae_encoder_create if encoder else self.dll.mugd_vae_create)(self.ctx, C.byref(c), C.byref(h)))
        n = Net(self, h, "vae_encoder" if encoder else "vae", dict(cfg))
        self._nets.append(n)
        return n

    def wave(self, cfg):
        c = WaveConfig()
        c.n_freq, c.middle_channels, c.num_res_blocks = cfg["n_freq"], cfg["middle_channels"], cfg["num_res_blocks"]
        c.num_heads, c.num_groups = cfg["num_heads"], cfg["num_groups"]
        c.n_levels = len(cfg["channel_mult"])
        _ilist(c.channel_mult, cfg["channel_mult"])
        c.n_attn = len(cfg["attention_resolutions"])
        _ilist(c.attention_resolutions, cfg["attention_resolutions"])
        h = _p()
        self.check(self.dll.mugd_wave_create(self.ctx, C.byref(c), C.byref(h)))
        n = Net(self, h, "wave", dict(cfg))
        self._nets.append(n)
        return n

    # ------------------------------------------------------------------ helpers
    def cond_embed(self, table, ids):
        table = self.f32(table)
        ids = ids.to(device=self.device, dtype=torch.int64).contiguous()
        B, ntok = ids.shape
        out = self.empty(B, table.shape[1], ntok)
        self.check(self.dll.mugd_cond_embed(self.ctx, _ptr(table), _ptr(ids), _ptr(out), B, ntok, table.shape[1]))
        return out

    def set_mel_pad_mode(self, mode):
        """'constant' / 0 (default): log_mel pads its centred frames with zeros (librosa >= 0.10); 'reflect' / 1: by reflection
        (librosa <= 0.9) -- include/mugd.h: mugd_set_mel_pad_mode."""
        refl = mode in (1, True, "reflect")
        if not refl and mode not in (0, False, "constant"):
            raise ValueError("pad mode must be 'constant' or 'reflect', got %r" % (mode,))
        self.check(self.dll.mugd_set_mel_pad_mode(self.ctx, 1 if refl else 0))
        self._mel_pad = 1 if refl else 0

    def log_mel(self, pcm, sr=22050, n_fft=512, hop=128, n_mels=128, pad_mode=None):
        """pad_mode: None = the context's setting (set_mel_pad_mode / MUGD_MEL_PAD), else 'constant' | 'reflect' for this call."""
        pcm = self.f32(pcm).reshape(-1)
        n = pcm.numel()
        out = self.empty(n_mels, 1 + n // hop)
        prev = getattr(self, "_mel_pad", 1 if os.environ.get("MUGD_MEL_PAD", "")[:1] in ("r", "1") else 0)
        if pad_mode is not None:
            self.set_mel_pad_mode(pad_mode)
        try:
            self.check(self.dll.mugd_log_mel(self.ctx, _ptr(pcm), n, sr, n_fft, hop, n_mels, _ptr(out)))
        finally:
            if pad_mode is not None:
                self.set_mel_pad_mode(prev)
        return out

    # ------------------------------------------------------------------ single operators
    def resample_poly(self, pcm, up, down):
        """scipy.signal.resample_poly(pcm, up, down) / librosa res_type="polyphase" on the device: mono fp32 -> fp32."""
        pcm = self.f32(pcm).reshape(-1)
        n_out = C.c_int64(0)
        self.check(self.dll.mugd_resample_poly(self.ctx, _p(None), pcm.numel(), int(up), int(down), _p(None), C.byref(n_out)))
        out = self.empty(n_out.value)
        self.check(self.dll.mugd_resample_poly(self.ctx, _ptr(pcm), pcm.numel(), int(up), int(down), _ptr(out), None))
        return out

    def timing_sweep(self, times, gap, offset, offset_is_f32, epsilon=10.0):
        """valid-note counts of (gap, offset) grid candidates (mug/data/utils.py:16-27): `times` is a float32 device
        tensor, the candidate arrays are NumPy (float64, float64, bool); returns an int32 NumPy array."""
        n_cand = len(gap)
        g = torch.from_numpy(np.ascontiguousarray(gap, dtype=np.float64)).to(self.device)
        o = torch.from_numpy(np.ascontiguousarray(offset, dtype=np.float64)).to(self.device)
        f = torch.from_numpy(np.ascontiguousarray(offset_is_f32, dtype=np.uint8)).to(self.device)
        counts = torch.empty(n_cand, dtype=torch.int32, device=self.device)
        self.check(self.dll.mugd_timing_sweep(self.ctx, _ptr(times), times.numel(), _ptr(g), _ptr(o), _ptr(f), n_cand,
                                              float(epsilon), _ptr(counts)))
        return counts.cpu().numpy()

    def remove_mini_jacks(self, start_ms, column, end_ms, jack_interval=90, column_width=128):
        """Host pass (mug/data/utils.py:140-255) on parsed arrays -> (new_x int32 with INT32_MIN = not moved, keep bool)."""
        n = len(start_ms)
        st = np.ascontiguousarray(start_ms, dtype=np.float64)
        co = np.ascontiguousarray(column, dtype=np.int32)
        en = np.ascontiguousarray(end_ms, dtype=np.float64)
        new_x = np.full(n, np.iinfo(np.int32).min, dtype=np.int32)
        keep = np.ones(n, dtype=np.uint8)
        rc = self.dll.mugd_remove_mini_jacks(n, st.ctypes.data_as(_p), co.ctypes.data_as(_p), en.ctypes.data_as(_p),
                                             float(jack_interval), int(column_width), new_x.ctypes.data_as(_p),
                                             keep.ctypes.data_as(_p))
        if rc != 0:
            raise MugdError("mugd_remove_mini_jacks failed with status %d" % rc)
        return new_x, keep.astype(bool)

    def op_group_norm(self, x, gamma, beta, groups, silu):
        x, gamma, beta = self.f32(x), self.f32(gamma), self.f32(beta)
        y = torch.empty_like(x)
        B, Cc, T = x.shape
        self.check(self.dll.mugd_op_group_norm(self.ctx, _ptr(x), _ptr(gamma), _ptr(beta), _ptr(y), B, Cc, T, groups, int(silu)))
        return y

    def op_layer_norm(self, x, gamma, beta):
        x, gamma, beta = self.f32(x), self.f32(gamma), self.f32(beta)
        y = torch.empty_like(x)
        B, Cc, T = x.shape
        self.check(self.dll.mugd_op_layer_norm(self.ctx, _ptr(x), _ptr(gamma), _ptr(beta), _ptr(y), B, Cc, T))
        return y

    def op_conv1d(self, x, w, bias=None, resid=None, dil=1, stride=1, pad=0, upsample=False, Tout=None, epi=0):
        x, w = self.f32(x), self.f32(w)
        if w.dim() == 2:
            w = w[:, :, None].contiguous()
        bias = self.f32(bias) if bias is not None else None
        resid = self.f32(resid) if resid is not None else None
        B, Cc, Tin = x.shape
        M, _, taps = w.shape
        if Tout is None:
            vin = 2 * Tin if upsample else Tin
            Tout = (vin + 2 * pad - dil * (taps - 1) - 1) // stride + 1
        y = self.empty(B, M // 2 if epi else M, Tout)
        self.check(self.dll.mugd_op_conv1d(self.ctx, _ptr(x), _ptr(w), _ptr(bias), _ptr(resid), _ptr(y),
                                           B, Cc, Tin, M, taps, dil, stride, pad, int(upsample), Tout, epi))
        return y

    def op_norm_conv1d(self, x, gamma, beta, w, bias=None, dil=1, pad=0, norm=1, groups=32, silu=False, wk=0):
        x, gamma, beta, w = self.f32(x), self.f32(gamma), self.f32(beta), self.f32(w)
        if w.dim() == 2:
            w = w[:, :, None].contiguous()
        bias = self.f32(bias) if bias is not None else None
        B, Cc, T = x.shape
        M, _, taps = w.shape
        y = self.empty(B, M, T + 2 * pad - dil * (taps - 1))
        self.check(self.dll.mugd_op_norm_conv1d(self.ctx, _ptr(x), _ptr(gamma), _ptr(beta), _ptr(w), _ptr(bias), _ptr(y),
                                                B, Cc, T, M, taps, dil, pad, norm, groups, int(silu), wk))
        return y

    def dev_bench_conv(self, B, Cc, T, M, taps=3, norm=0, gated=False, wk=0, tn=0, copies=1, iters=50):
        us = _f()
        self.check(self.dll.mugd_dev_bench_conv(self.ctx, B, Cc, T, M, taps, norm, int(gated), wk, tn, copies, iters, C.byref(us)))
        return us.value

    def dev_clock_probe(self):
        """Shader clock (MHz) sustained under matrix load: which class of box a measurement ran on."""
        mhz = _f()
        self.check(self.dll.mugd_dev_clock_probe(self.ctx, C.byref(mhz)))
        return mhz.value

    def op_attention(self, q, k, v, rel, cemb, heads):
        q, k, v, rel, cemb = map(self.f32, (q, k, v, rel, cemb))
        B, Cc, Tq = q.shape
        Tk = k.shape[2]
        out = torch.empty_like(q)
        self.check(self.dll.mugd_op_attention(self.ctx, _ptr(q), _ptr(k), _ptr(v), _ptr(rel), _ptr(cemb), _ptr(out),
                                              B, heads, Cc // heads, Tq, Tk, (rel.shape[0] - 1) // 2))
        return out

    def op_s4_kernel(self, Cp, Bp, Pp, inv_w_real, w_imag, log_dt, Lint, L):
        Cp, Bp, Pp, inv_w_real, w_imag, log_dt = map(self.f32, (Cp, Bp, Pp, inv_w_real, w_imag, log_dt))
        H, N = inv_w_real.shape
        k = self.empty(H, L)
        self.check(self.dll.mugd_op_s4_kernel(self.ctx, _ptr(Cp), _ptr(Bp), _ptr(Pp), _ptr(inv_w_real), _ptr(w_imag),
                                              _ptr(log_dt), _ptr(k), H, N, Lint, L))
        return k

    def op_s4_conv(self, u, k, D):
        u, k, D = self.f32(u), self.f32(k), self.f32(D)
        B, H, L = u.shape
        y = torch.empty_like(u)
        self.check(self.dll.mugd_op_s4_conv(self.ctx, _ptr(u), _ptr(k), _ptr(D), _ptr(y), B, H, L))
        return y

    def op_gn_s4_conv(self, u, k, D, gamma, beta, groups):
        u, k, D, gamma, beta = map(self.f32, (u, k, D, gamma, beta))
        B, H, L = u.shape
        y = torch.empty_like(u)
        self.check(self.dll.mugd_op_gn_s4_conv(self.ctx, _ptr(u), _ptr(k), _ptr(D), _ptr(gamma), _ptr(beta), groups, _ptr(y), B, H, L))
        return y

    # ------------------------------------------------------------------ training slice (include/mugd.h)
    def train_q_sample(self, x0, noise, t, sqrt_ac, sqrt_1mac):
        """diffusion.py:326-333."""
        x0, noise, sqrt_ac, sqrt_1mac = map(self.f32, (x0, noise, sqrt_ac, sqrt_1mac))
        t = t.to(device=self.device, dtype=torch.int64).contiguous()
        out = torch.empty_like(x0)
        B = x0.shape[0]
        self.check(self.dll.mugd_train_q_sample(self.ctx, _ptr(x0), _ptr(noise), _ptr(t), _ptr(sqrt_ac), _ptr(sqrt_1mac), _ptr(out), B, x0.numel() // B))
        return out

    def train_smooth_l1(self, pred, target, beta=0.02, add=0.01, want_grad=True):
        """diffusion.py:341-354,386: per-sample loss (B) and d(mean loss)/d pred."""
        pred, target = self.f32(pred), self.f32(target)
        B = pred.shape[0]
        loss = self.empty(B)
        grad = torch.empty_like(pred) if want_grad else None
        self.check(self.dll.mugd_train_smooth_l1(self.ctx, _ptr(pred), _ptr(target), float(beta), float(add), _ptr(loss), _ptr(grad), B, pred.numel() // B))
        return loss, grad

    _RESBLOCK_KEYS = dict(gn1_w="in_layers.0.weight", gn1_b="in_layers.0.bias", conv1_w="in_layers.2.weight", conv1_b="in_layers.2.bias",
                          emb_w="emb_layers.1.weight", emb_b="emb_layers.1.bias", gn2_w="out_layers.0.weight", gn2_b="out_layers.0.bias",
                          conv2_w="out_layers.3.weight", conv2_b="out_layers.3.bias", skip_w="skip_connection.weight", skip_b="skip_connection.bias")
    _RESNET_KEYS = dict(gn1_w="norm1.weight", gn1_b="norm1.bias", conv1_w="conv1.weight", conv1_b="conv1.bias", gn2_w="norm2.weight",
                        gn2_b="norm2.bias", conv2_w="conv2.weight", conv2_b="conv2.bias", skip_w="nin_shortcut.weight", skip_b="nin_shortcut.bias")

    # Argument packs.  A block's parameter pointers (and, with a persistent gradient arena, its gradient pointers) do not change from
    # step to step (AdamW updates in place): a caller that keeps a `pack` dict per block pays for building the ctypes arrays and the
    # tensor views ONCE; without a pack everything is rebuilt per call.
    def _resblock_ptrs(self, keymap, params, want_grads, pack=None):
        if pack is not None and "P" in pack and (not want_grads or "G" in pack):
            return pack["P"], pack.get("G") or ResBlockPtrs(), None, (pack["grads"] if want_grads else {})
        P, keep = ResBlockPtrs(), []
        ts = {}
        for f, k in keymap.items():
            if k in params:
                t = self.f32(params[k])
                keep.append(t)
                ts[f] = (k, t)
                setattr(P, f, t.data_ptr())
        G, grads = ResBlockPtrs(), {}
        if want_grads:
            for f, (k, t) in ts.items():
                g = self.zgrad(t)
                keep.append(g)
                grads[k] = g
                setattr(G, f, g.data_ptr())
        if pack is not None:
            pack["P"] = P
            pack.setdefault("keep", []).extend(keep)
            if want_grads:
                pack["G"], pack["grads"] = G, grads
        return P, G, keep, grads

    def _array_ptrs(self, keys, params, want_grads, pack=None):
        """(pointer array of the parameters in `keys` order, pointer array of their gradients, keep-alive list, grads dict)."""
        n = len(keys)
        if pack is not None and "P" in pack and (not want_grads or "G" in pack):
            return pack["P"], pack.get("G") or (C.c_void_p * n)(), None, (pack["grads"] if want_grads else {})
        keep, grads = [], {}
        PA, GA = (C.c_void_p * n)(), (C.c_void_p * n)()
        for i, k in enumerate(keys):
            t = self.f32(params[k])
            keep.append(t)
            PA[i] = t.data_ptr()
            if want_grads:
                g = self.zgrad(t)
                keep.append(g)
                grads[k] = g
                GA[i] = g.data_ptr()
        if pack is not None:
            pack["P"] = PA
            pack.setdefault("keep", []).extend(keep)
            if want_grads:
                pack["G"], pack["grads"] = GA, grads
        return PA, GA, keep, grads

    def train_release_states(self):
        self.check(self.dll.mugd_train_release_states(self.ctx))

    def train_set_precision(self, bf16):
        """Training GEMMs on the bf16 matrix cores (fp32 accumulation; BASELINE configs[4]) instead of the fp32-input MFMA parity mode."""
        self.check(self.dll.mugd_train_set_precision(self.ctx, 1 if bf16 else 0))

    def train_step_begin(self):
        """Open the step bracket (include/mugd.h): weights must not change until train_step_end; packed bf16 weights come from a cache
        refreshed here by one launch, weight / bias gradients are complete after train_step_flush / train_step_end."""
        self.check(self.dll.mugd_train_step_begin(self.ctx))

    def train_step_flush(self):
        self.check(self.dll.mugd_train_step_flush(self.ctx))

    def train_step_end(self):
        self.check(self.dll.mugd_train_step_end(self.ctx))

    # The bracket's packed-weight cache is keyed by tensor address and re-read by the next train_step_begin (include/mugd.h): the binding
    # keeps whatever owns those tensors (a mug.train.TrainPlan: its state dict and argument packs) referenced until the cache is dropped.
    _bracket_owner = None

    def train_step_reset(self, owner=None):
        """Drop the library's packed-weight cache (no pointer into parameter tensors survives) and hand the bracket to `owner`."""
        self.check(self.dll.mugd_train_step_reset(self.ctx))
        self._bracket_owner = owner

    def train_profile(self, enable):
        """Event-bracket the training GEMM launches (True), or stop and return {'conv': {ms, flops, launches}, 'wgrad': {...}} (False)."""
        if enable:
            self.check(self.dll.mugd_train_profile(self.ctx, 1, None))
            return None
        out = (C.c_double * 6)()
        self.check(self.dll.mugd_train_profile(self.ctx, 0, out))
        return {"conv": dict(ms=out[0], flops=out[2], launches=int(out[4])), "wgrad": dict(ms=out[1], flops=out[3], launches=int(out[5]))}

    def train_concat(self, a, b):
        """cat([a, b], dim=1) of (B, C, T) tensors (skip / audio concatenation)."""
        a, b = self.f32(a), self.f32(b)
        B, Ca, T = a.shape
        out = self.empty(B, Ca + b.shape[1], T)
        self.check(self.dll.mugd_train_concat(self.ctx, _ptr(a), _ptr(b), _ptr(out), B, Ca, b.shape[1], T))
        return out

    def train_split(self, src, Ca, acc_a=None, acc_b=None):
        """(src[:, :Ca], src[:, Ca:]) as contiguous tensors; acc_a / acc_b: existing tensors the slice is ADDED to instead."""
        src = self.f32(src)
        B, Ct, T = src.shape
        a = acc_a if acc_a is not None else self.empty(B, Ca, T)
        b = acc_b if acc_b is not None else self.empty(B, Ct - Ca, T)
        self.check(self.dll.mugd_train_split(self.ctx, _ptr(src), _ptr(a), _ptr(b), B, Ca, Ct - Ca, T, int(acc_a is not None), int(acc_b is not None)))
        return a, b

    def train_add(self, a, b, out=None):
        """a + b natively (out may be a or b)."""
        out = torch.empty_like(a) if out is None else out
        self.check(self.dll.mugd_train_add(self.ctx, _ptr(a), _ptr(b), _ptr(out), a.numel()))
        return out

    def train_resblock(self, params, x, emb, dy, groups=32, state=None, pack=None):
        """TimestepResBlock forward + backward (unet.py:212-239).  params: dict with the module's tensors
        (in_layers.0.weight/bias, in_layers.2.weight/bias, emb_layers.1.weight/bias, out_layers.0.weight/bias,
        out_layers.3.weight/bias, optionally skip_connection.weight/bias).  Returns y, dx, demb, grads (same keys);
        dy None: forward only (dx, demb None, grads empty)."""
        x, emb = self.f32(x), self.f32(emb)
        dy = None if dy is None else self.f32(dy)
        P, G, keep, grads = self._resblock_ptrs(self._RESBLOCK_KEYS, params, dy is not None, pack)
        B, Cin, T = x.shape
        Cout = params["in_layers.2.weight"].shape[0]
        y = self.empty(B, Cout, T)
        dx = None if dy is None else torch.empty_like(x)
        demb = None if dy is None else torch.empty_like(emb)
        self.check(self.dll.mugd_train_resblock(self.ctx, C.byref(P), _ptr(x), _ptr(emb), _ptr(dy), _ptr(y), _ptr(dx), _ptr(demb), C.byref(G),
                                                B, Cin, Cout, T, emb.shape[1], groups, _sref(state)))
        return y, dx, demb, grads

    def train_resnet_block(self, params, x, dy, groups=32, dilations=(1, 1), state=None, pack=None):
        """ResnetBlock forward + backward (mug/model/models.py:142-159: norm1/conv1/norm2/conv2/nin_shortcut, dilated convs)."""
        x = self.f32(x)
        dy = None if dy is None else self.f32(dy)
        P, G, keep, grads = self._resblock_ptrs(self._RESNET_KEYS, params, dy is not None, pack)
        B, Cin, T = x.shape
        Cout = params["conv1.weight"].shape[0]
        y = self.empty(B, Cout, T)
        dx = None if dy is None else torch.empty_like(x)
        self.check(self.dll.mugd_train_resnet_block(self.ctx, C.byref(P), _ptr(x), _ptr(dy), _ptr(y), _ptr(dx), C.byref(G), B, Cin, Cout, T, groups,
                                                    int(dilations[0]), int(dilations[1]), _sref(state)))
        return y, dx, grads

    def train_time_embed(self, params, temb, demb, pack=None):
        """time_embed (unet.py:334-339): params '0.weight', '0.bias', '2.weight', '2.bias'; temb (B, K).  Returns emb, grads.
        pack (a dict kept by the caller across steps): the four gradient tensors are taken ONCE and reused, like every block's."""
        temb = self.f32(temb)
        demb = None if demb is None else self.f32(demb)
        w1, b1, w2, b2 = (self.f32(params[k]) for k in ("0.weight", "0.bias", "2.weight", "2.bias"))
        B, K = temb.shape
        M = w1.shape[0]
        emb = self.empty(B, M)
        if demb is None:
            g = [None] * 4
        elif pack is not None and "g" in pack:
            g = pack["g"]
        else:
            g = [self.zgrad(t) for t in (w1, b1, w2, b2)]
            if pack is not None:
                pack["g"] = g
        self.check(self.dll.mugd_train_time_embed(self.ctx, _ptr(w1), _ptr(b1), _ptr(w2), _ptr(b2), _ptr(temb), _ptr(demb), _ptr(emb),
                                                  _ptr(g[0]), _ptr(g[1]), _ptr(g[2]), _ptr(g[3]), B, K, M))
        return emb, ({} if demb is None else {"0.weight": g[0], "0.bias": g[1], "2.weight": g[2], "2.bias": g[3]})

    def train_embedding_bwd(self, ids, dcontext, rows, pack=None):
        """BeatmapFeatureEmbedder backward: gradient of the (rows, dim) table from dcontext (B, dim, ntok).
        pack (a dict kept by the caller across steps): the gradient tensor is allocated ONCE and rewritten every step."""
        ids = ids.to(device=self.device, dtype=torch.int64).contiguous()
        dc = self.f32(dcontext)
        B, dim, ntok = dc.shape
        if pack is not None and "dt" in pack:
            dt = pack["dt"]
        else:
            dt = self.empty(rows, dim)      # every row is written by the kernel
            if pack is not None:
                pack["dt"] = dt
        self.check(self.dll.mugd_train_embedding_bwd(self.ctx, _ptr(ids), _ptr(dc), _ptr(dt), B, ntok, dim, rows))
        return dt

    def train_conv(self, weight, bias, x, dy, dil=1, mode=0, gn=None, groups=32, state=None, pack=None):
        """conv1d forward + backward; mode 0 plain (padding = dil (k - 1) / 2), 1 Downsample, 2 Upsample (models.py:55-91);
        gn = (weight, bias): GroupNorm + SiLU in front (the U-Net's out head).  Returns y, dx, dw, db, (dgn_w, dgn_b) or None."""
        x = self.f32(x)
        dy = None if dy is None else self.f32(dy)
        if pack is not None and "w" in pack:
            w, b, gw, gb = pack["w"], pack["b"], pack["gw"], pack["gb"]
        else:
            w = self.f32(weight)
            b = None if bias is None else self.f32(bias)
            gw, gb = (self.f32(gn[0]), self.f32(gn[1])) if gn is not None else (None, None)
            if pack is not None:
                pack.update(w=w, b=b, gw=gw, gb=gb)
        B, Cin, Tin = x.shape
        Cout, _, taps = w.shape
        Tout = Tin // 2 if mode == 1 else (2 * Tin if mode == 2 else Tin)
        y = self.empty(B, Cout, Tout)
        dx = dw = db = dgw = dgb = None
        if dy is not None:
            dx = torch.empty_like(x)
            if pack is not None and "dw" in pack:
                dw, db, dgw, dgb = pack["dw"], pack["db"], pack["dgw"], pack["dgb"]
            else:
                dw = self.zgrad(w)
                db = None if b is None else self.zgrad(b)
                if gn is not None:
                    dgw, dgb = self.zgrad(gw), self.zgrad(gb)
                if pack is not None:
                    pack.update(dw=dw, db=db, dgw=dgw, dgb=dgb)
        self.check(self.dll.mugd_train_conv(self.ctx, _ptr(w), _ptr(b), _ptr(gw), _ptr(gb), _ptr(x), _ptr(dy), _ptr(y), _ptr(dx), _ptr(dw), _ptr(db),
                                            _ptr(dgw), _ptr(dgb), B, Cin, Cout, Tin, taps, int(dil), int(mode), int(groups), _sref(state)))
        return y, dx, dw, db, (None if (gn is None or dy is None) else (dgw, dgb))

    # include/mugd.h MUGD_S4_*: the S4Layer's tensors in the C ABI's order, by their state-dict names
    S4LAYER_KEYS = ("norm.weight", "norm.bias", "s4_model.kernel.kernel.C", "s4_model.kernel.kernel.B", "s4_model.kernel.kernel.P",
                    "s4_model.kernel.kernel.inv_w_real", "s4_model.kernel.kernel.w_imag", "s4_model.kernel.kernel.log_dt", "s4_model.D",
                    "s4_model.output_linear.0.weight", "s4_model.output_linear.0.bias", "out_layer.weight", "out_layer.bias")

    def train_s4layer(self, params, x, dy, groups=32, state=None, pack=None):
        """S4Layer forward + backward (unet.py:76-91, s4.py:1471-1541, kernel gradients included).  params: dict keyed like the module's
        state dict (S4LAYER_KEYS + 's4_model.kernel.kernel.L', the stored internal length).  Returns y, dx, grads."""
        x = self.f32(x)
        dy = None if dy is None else self.f32(dy)
        PA, GA, keep, grads = self._array_ptrs(self.S4LAYER_KEYS, params, dy is not None, pack)
        B, H, T = x.shape
        N = params["s4_model.kernel.kernel.inv_w_real"].shape[-1]
        if pack is not None and "Lint" in pack:
            Lint = pack["Lint"]
        else:
            Lint = int(params["s4_model.kernel.kernel.L"])          # a device-resident buffer costs a host synchronisation here: packs keep the value
            if pack is not None:
                pack["Lint"] = Lint
        y, dx = torch.empty_like(x), (None if dy is None else torch.empty_like(x))
        self.check(self.dll.mugd_train_s4layer(self.ctx, PA, _ptr(x), _ptr(dy), _ptr(y), _ptr(dx), GA, B, H, T, N, Lint, int(groups), _sref(state)))
        return y, dx, grads

    # include/mugd.h MUGD_TF_*: the ContextualTransformer's tensors in the C ABI's order, by their state-dict names
    TRANSFORMER_KEYS = (
        "norm.weight", "norm.bias", "proj_in.weight", "proj_in.bias",
        "transformer_blocks.0.norm1.weight", "transformer_blocks.0.norm1.bias",
        "transformer_blocks.0.attn1.to_q.weight", "transformer_blocks.0.attn1.to_k.weight", "transformer_blocks.0.attn1.to_v.weight",
        "transformer_blocks.0.attn1.to_out.0.weight", "transformer_blocks.0.attn1.to_out.0.bias",
        "transformer_blocks.0.attn1.relative_position_embedding", "transformer_blocks.0.attn1.C_embedding",
        "transformer_blocks.0.norm2.weight", "transformer_blocks.0.norm2.bias",
        "transformer_blocks.0.attn2.to_q.weight", "transformer_blocks.0.attn2.to_k.weight", "transformer_blocks.0.attn2.to_v.weight",
        "transformer_blocks.0.attn2.to_out.0.weight", "transformer_blocks.0.attn2.to_out.0.bias",
        "transformer_blocks.0.attn2.relative_position_embedding", "transformer_blocks.0.attn2.C_embedding",
        "transformer_blocks.0.norm3.weight", "transformer_blocks.0.norm3.bias",
        "transformer_blocks.0.ff.net.0.proj.weight", "transformer_blocks.0.ff.net.0.proj.bias",
        "transformer_blocks.0.ff.net.2.weight", "transformer_blocks.0.ff.net.2.bias", "proj_out.weight", "proj_out.bias")

    def train_transformer(self, params, x, context, dy, heads, groups=32, state=None, pack=None):
        """ContextualTransformer forward + backward (mug/model/attention.py:154-199).  params: dict keyed like the module's state dict
        (TRANSFORMER_KEYS).  x, dy (B, C, T); context (B, Cc, Tk) or None (attn2 = second self-attention).
        Returns y, dx, dcontext (None without context), grads (same keys)."""
        x = self.f32(x)
        dy = None if dy is None else self.f32(dy)
        ctx = None if context is None else self.f32(context)
        PA, GA, keep, grads = self._array_ptrs(self.TRANSFORMER_KEYS, params, dy is not None, pack)
        B, Cm, T = x.shape
        pmax = (params["transformer_blocks.0.attn1.relative_position_embedding"].shape[0] - 1) // 2
        y, dx = torch.empty_like(x), (None if dy is None else torch.empty_like(x))
        dctx = None if (ctx is None or dy is None) else torch.empty_like(ctx)
        Cc, Tk = (0, 0) if ctx is None else (ctx.shape[1], ctx.shape[2])
        self.check(self.dll.mugd_train_transformer(self.ctx, PA, _ptr(x), _ptr(ctx), _ptr(dy), _ptr(y), _ptr(dx), _ptr(dctx), GA,
                                                   B, Cm, T, Cc, Tk, int(heads), int(groups), int(pmax), _sref(state)))
        return y, dx, dctx, grads

    def train_adamw(self, param, grad, exp_avg, exp_avg_sq, step, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.01):
        """One torch.optim.AdamW step, in place on device tensors."""
        self.check(self.dll.mugd_train_adamw(self.ctx, _ptr(param), _ptr(grad), _ptr(exp_avg), _ptr(exp_avg_sq), param.numel(), float(lr),
                                             float(betas[0]), float(betas[1]), float(eps), float(weight_decay), int(step)))

    def train_adamw_multi(self, params, grads, exp_avg, exp_avg_sq, step, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.01):
        """One AdamW step over lists of device tensors (in place), a single native call."""
        n = len(params)
        arr = lambda ts: (C.c_void_p * n)(*[t.data_ptr() for t in ts])          # noqa: E731
        sizes = (C.c_longlong * n)(*[p.numel() for p in params])
        self.check(self.dll.mugd_train_adamw_multi(self.ctx, n, arr(params), arr(grads), arr(exp_avg), arr(exp_avg_sq), sizes, float(lr),
                                                   float(betas[0]), float(betas[1]), float(eps), float(weight_decay), int(step)))

    def op_timestep_embedding(self, t, dim):
        t = t.to(device=self.device, dtype=torch.int64).contiguous()
        out = self.empty(t.shape[0], dim)
        self.check(self.dll.mugd_op_timestep_embedding(self.ctx, _ptr(t), _ptr(out), t.shape[0], dim))
        return out


class Net:
    """A native network instance (U-Net / VAE decoder / wave encoder)."""

    def __init__(self, lib, handle, kind, cfg):
        self.lib, self.h, self.kind, self.cfg = lib, handle, kind, cfg
        self._keep = {}      # name -> tensor: keeps the borrowed device memory alive

    def close(self):
        if self.h:
            self.lib.dll.mugd_net_destroy(self.h)
            self.h = _p()

    def set_params(self, tensors, prefix=""):
        """tensors: name -> torch tensor.  Names starting with `prefix` are registered (prefix stripped)."""
        lib = self.lib
        for name, t in tensors.items():
            if prefix and not name.startswith(prefix):
                continue
            key = name[len(prefix):]
            if t.dtype == torch.int64:
                td, code = t.detach().to(lib.device).contiguous(), I64
            else:
                td, code = lib.f32(t), F32
            self._keep[key] = td
            shape = (C.c_int64 * max(1, td.dim()))(*td.shape)
            lib.check(lib.dll.mugd_net_set_param(self.h, key.encode(), _ptr(td), code, td.dim(), shape))
        lib.check(lib.dll.mugd_net_invalidate(self.h))

    # U-Net ------------------------------------------------------------------
    def _audio_ptrs(self, audio, B):
        """audio: the wave-encoder maps (the last n_levels are used); their batch size may be any
        divisor of B (seeds sharing one audio share one copy of its features)."""
        nl = len(self.cfg["channel_mult"])
        maps = [self.lib.f32(a) for a in list(audio)[-nl:]]
        ab = maps[0].shape[0]
        if any(m.shape[0] != ab for m in maps) or B % ab != 0:
            raise MugdError("audio maps must share a batch size that divides %d" % B)
        arr = (_p * nl)(*[_p(a.data_ptr()) for a in maps])
        return maps, arr, ab

    def forward(self, x, t, context, audio):
        lib = self.lib
        x, context = lib.f32(x), lib.f32(context)
        t = t.to(device=lib.device, dtype=torch.int64).contiguous()
        B, _, z = x.shape
        maps, arr, ab = self._audio_ptrs(audio, B)
        eps = lib.empty(B, self.cfg["out_channels"], z)
        lib.check(lib.dll.mugd_unet_forward(self.h, _ptr(x), _ptr(t), _ptr(context), context.shape[2], arr, ab, _ptr(eps), B, z))
        return eps

    def profile(self):
        """Per-kernel-class {ms, flops, launches} of one eager pass of the last compiled program."""
        ms = (C.c_double * PROFILE_KINDS)()
        fl = (C.c_double * PROFILE_KINDS)()
        ln = (C.c_int64 * PROFILE_KINDS)()
        self.lib.check(self.lib.dll.mugd_net_profile(self.h, ms, fl, ln))
        return {self.lib.dll.mugd_profile_kind_name(k).decode(): dict(ms=ms[k], flops=fl[k], launches=int(ln[k]))
                for k in range(PROFILE_KINDS)}

    def host_enqueue(self, passes=5):
        """(host microseconds, ops) per enqueue of the last compiled program (include/mugd.h: mugd_net_host_enqueue)."""
        us, n = C.c_double(0), C.c_int64(0)
        self.lib.check(self.lib.dll.mugd_net_host_enqueue(self.h, int(passes), C.byref(us), C.byref(n)))
        return us.value, int(n.value)

    def ddim_sample(self, x_T, c, audio, timesteps, sched, uc=None, scale=1.0, noise=None, want_pred_x0=False, want_first=False):
        """timesteps: sequence of ints in sampling order; sched: (S,4) float32 rows {a_t, a_prev, sigma, sqrt(1-a_t)}."""
        lib = self.lib
        x = lib.f32(x_T).clone()
        c = lib.f32(c)
        uc = lib.f32(uc) if uc is not None else None
        B, _, z = x.shape
        maps, arr, ab = self._audio_ptrs(audio, B)
        S = len(timesteps)
        ts = (C.c_int64 * S)(*[int(v) for v in timesteps])
        flat = [float(v) for row in sched for v in row]
        sc = (_f * (4 * S))(*flat)
        noise = lib.f32(noise) if noise is not None else None
        pred = torch.empty_like(x) if want_pred_x0 else None
        first = torch.empty((2,) + tuple(x.shape), dtype=x.dtype, device=x.device) if want_first else None
        lib.check(lib.dll.mugd_ddim_sample(self.h, _ptr(x), _ptr(c), _ptr(uc), c.shape[2], arr, ab, B, z, S, ts, sc,
                                           float(scale), _ptr(noise), _ptr(pred), _ptr(first)))
        if want_first:
            return x, pred, first
        return (x, pred) if want_pred_x0 else x

    # VAE --------------------------------------------------------------------
    def decode(self, z_lat):
        lib = self.lib
        z_lat = lib.f32(z_lat)
        B, _, z = z_lat.shape
        up = 2 ** (len(self.cfg["channel_mult"]) - 1)
        out = lib.empty(B, self.cfg["x_channels"], z * up)
        lib.check(lib.dll.mugd_vae_decode(self.h, _ptr(z_lat), _ptr(out), B, z))
        return out

    def vae_encode(self, x):
        """Encoder moments (B, 2 z_ch, T / 2^(levels-1)) of a note-grid tensor x (B, x_ch, T)."""
        lib = self.lib
        x = lib.f32(x)
        B, _, T = x.shape
        down = 2 ** (len(self.cfg["channel_mult"]) - 1)
        out = lib.empty(B, 2 * self.cfg["z_channels"], T // down)
        lib.check(lib.dll.mugd_vae_encode(self.h, _ptr(x), _ptr(out), B, T))
        return out

    # wave encoder -------------------------------------------------------------
    def encode(self, mel, only_last=None):
        lib = self.lib
        mel = lib.f32(mel)
        B, _, Ta = mel.shape
        mults = self.cfg["channel_mult"]
        nl = len(mults)
        outs = []
        for l in range(nl):
            if only_last is not None and l < nl - only_last:
                outs.append(None)
            else:
                outs.append(lib.empty(B, self.cfg["middle_channels"] * mults[l], Ta >> l))
        arr = (_p * nl)(*[_ptr(o) for o in outs])
        lib.check(lib.dll.mugd_wave_encode(self.h, _ptr(mel), arr, B, Ta))
        return outs


_default = None


def _build_in_tree():
    """libmugd.so is a build product (git-ignored): on a checkout where nobody ran the build yet, compile it now with hipcc
    (about a minute).  This is still the HIP path -- without hipcc the loader raises as before."""
    import importlib.util
    import sys
    spec = importlib.util.spec_from_file_location("mugd_build", os.path.join(os.path.dirname(_HERE), "build.py"))
    b = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(b)
    # one process per GPU (mug/shard.py): every rank lands here at once on a fresh checkout -- serialise the build with a
    # file lock and re-check under it; build.py links to a temporary name and renames it into place
    import fcntl
    with open(LIB_PATH + ".lock", "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            if not os.path.exists(LIB_PATH):
                print("mug._native: %s is missing, building it with hipcc ..." % LIB_PATH, file=sys.stderr, flush=True)
                b.build(verbose=False)
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)


def get_lib():
    """The process-wide library instance on the current GPU.  Raises if the HIP build or a GPU is missing."""
    global _default
    if _default is None:
        if not os.path.exists(LIB_PATH) and torch.cuda.is_available():
            try:
                _build_in_tree()
            except Exception as e:          # no hipcc / compile error: Lib() below reports the missing library
                raise MugdError("libmugd.so is missing and could not be built: %s" % e) from e
        _default = Lib()
    return _default
