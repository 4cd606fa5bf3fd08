"""Training (BASELINE configs[4]).  What exists natively (include/mugd.h, "training
slice"): q_sample, the smooth-L1 noise-prediction loss with its gradient, forward + backward of TimestepResBlock and of
ContextualTransformer (`Lib.train_resblock`, `Lib.train_transformer`), an AdamW step; here: the glue that turns them into a
data-parallel step on a block and the gradient all-reduce every rank of a DDP job runs (one flat bucket, averaged).  The backward
of the S4 and resampling layers -- and with them `DDPM.training_step` (mug/diffusion/diffusion.py:356-414) for the whole U-Net --
are built from these block entry points by `training_step` below: a forward sweep that keeps every block's INPUT, then a backward
sweep that calls each block again with its upstream gradient.  By default the block keeps its forward intermediates between the
two calls (a `TrainState`); with recompute=True it recomputes its own forward in the backward call instead (block-level
checkpointing: 2 forwards + 1 backward per step, no whole-network activation storage).  All arithmetic is native, including the
channel concatenation / slicing of the skip and audio connections and the gradient accumulation over them (mugd_train_concat /
_split / _add); torch owns the tensors.  No entry point synchronises the host: a step is one uninterrupted stream of kernels.
`lib.train_set_precision(True)` runs the GEMMs on the bf16 matrix cores (BASELINE configs[4]); the default is the fp32 parity mode.
"""
import torch
import torch.distributed as dist


def resblock_loss_and_grads(lib, params, x, emb, target, groups=32, beta=0.02, add=0.01):
    """loss = mean_b( mean_{c,t} smooth_l1(target, block(x, emb); beta) + add )  and its gradients w.r.t. the block's parameters
    (diffusion.py:341-354,386 applied to one block's output).  Two native calls: forward for the prediction, then forward + backward
    with the loss gradient as upstream."""
    zero = torch.zeros((x.shape[0], params["in_layers.2.weight"].shape[0], x.shape[2]), dtype=torch.float32)
    y, _, _, _ = lib.train_resblock(params, x, emb, zero, groups=groups)
    loss, dy = lib.train_smooth_l1(y, target, beta=beta, add=add)
    _, _, _, grads = lib.train_resblock(params, x, emb, dy, groups=groups)
    return loss.mean(), grads


def allreduce_gradients(grads, average=True, group=None):
    """One bucketed all_reduce over every gradient tensor (a flat fp32 buffer: a single collective per step; RCCL over xGMI with
    the `nccl` backend, `gloo` in the CPU tests), written back in place.  With equal shard sizes and a per-rank mean loss, the
    averaged gradients equal the full-batch gradient."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return grads
    keys = sorted(grads)
    dev = grads[keys[0]].device
    backend_dev = dev if dist.get_backend(group) == "nccl" else torch.device("cpu")
    flat = torch.cat([grads[k].reshape(-1).to(backend_dev) for k in keys])
    dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
    if average:
        flat /= dist.get_world_size(group)
    off = 0
    for k in keys:
        n = grads[k].numel()
        grads[k].copy_(flat[off:off + n].reshape(grads[k].shape).to(dev))
        off += n
    return grads


def broadcast_parameters(sd, src=0, group=None, bucket_bytes=64 << 20):
    """What DistributedDataParallel does at construction: every rank takes rank `src`'s trainable tensors (flat buckets, in place),
    so data-parallel training cannot start from silently different initialisations.  No-op without an initialised process group."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return
    keys = [k for k, v in sd.items() if v.dtype == torch.float32 and k.startswith("model.")]
    nccl = dist.get_backend(group) == "nccl"
    i = 0
    while i < len(keys):
        chunk, nbytes = [], 0
        while i < len(keys) and (not chunk or nbytes < bucket_bytes):
            chunk.append(keys[i]); nbytes += sd[keys[i]].numel() * 4; i += 1
        dev = sd[chunk[0]].device if nccl else torch.device("cpu")
        flat = torch.cat([sd[k].reshape(-1).to(dev) for k in chunk])
        dist.broadcast(flat, src=src, group=group)
        off = 0
        for k in chunk:
            n = sd[k].numel()
            sd[k].copy_(flat[off:off + n].reshape(sd[k].shape).to(sd[k].device))
            off += n


def adamw_step(lib, params, grads, state, step, lr=1e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.01):
    """torch.optim.AdamW semantics on a parameter dict, in place on device tensors; `state` holds exp_avg / exp_avg_sq.  One native
    call for the whole list."""
    keys = list(params)
    for k in keys:
        if k not in state:
            state[k] = (torch.zeros_like(params[k]), torch.zeros_like(params[k]))
        assert params[k].is_contiguous() and params[k].dtype == torch.float32 and params[k].device == lib.device, k
    gs = [lib.f32(grads[k]) for k in keys]
    lib.train_adamw_multi([params[k] for k in keys], gs, [state[k][0] for k in keys], [state[k][1] for k in keys], step,
                          lr=lr, betas=betas, eps=eps, weight_decay=weight_decay)
    return params


class AdamW:
    """adamw_step with its argument lists built once: for a TrainPlan's stable (parameter, gradient) tensors one optimiser step is a
    single native call with prebuilt pointer arrays (configure_optimizers, diffusion.py:477-499: AdamW over the trainable tensors)."""

    def __init__(self, lib, params, grads, lr=1e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.01):
        self.lib, self.hp = lib, (float(lr), float(betas[0]), float(betas[1]), float(eps), float(weight_decay))
        self.keys = list(grads)
        self.tensors = [(params[k], grads[k], torch.zeros_like(params[k]), torch.zeros_like(params[k])) for k in self.keys]
        for p_, g_, _, _ in self.tensors:
            assert p_.is_contiguous() and g_.is_contiguous() and p_.dtype == torch.float32 and p_.device == lib.device and p_.shape == g_.shape
        # every tensor cut into runs of <= 4096 elements: one descriptor row {param, grad, exp_avg, exp_avg_sq address, count} per
        # run, one workgroup per row, ONE launch per optimiser step (mugd_train_adamw_chunks)
        import numpy as np
        rows = []
        for p_, g_, m_, v_ in self.tensors:
            n = p_.numel()
            offs = np.arange(0, n, 4096, dtype=np.int64)
            cnt = np.minimum(4096, n - offs)
            rows.append(np.stack([p_.data_ptr() + 4 * offs, g_.data_ptr() + 4 * offs, m_.data_ptr() + 4 * offs, v_.data_ptr() + 4 * offs, cnt], axis=1))
        desc = np.concatenate(rows).astype(np.int64)
        self.nchunks = int(desc.shape[0])
        self.desc = torch.from_numpy(desc).to(lib.device)
        self.gptrs = [g_.data_ptr() for _, g_, _, _ in self.tensors]
        self.t = 0

    def _rebind(self, grads):
        """Point the descriptor table at `grads` (same keys / shapes) -- keeps the moment tensors and the step count."""
        import numpy as np
        self.tensors = [(p_, grads[k], m_, v_) for k, (p_, _, m_, v_) in zip(self.keys, self.tensors)]
        rows = []
        for p_, g_, m_, v_ in self.tensors:
            assert g_.is_contiguous() and g_.dtype == torch.float32 and g_.device == p_.device and g_.shape == p_.shape
            n = p_.numel()
            offs = np.arange(0, n, 4096, dtype=np.int64)
            cnt = np.minimum(4096, n - offs)
            rows.append(np.stack([p_.data_ptr() + 4 * offs, g_.data_ptr() + 4 * offs, m_.data_ptr() + 4 * offs, v_.data_ptr() + 4 * offs, cnt], axis=1))
        self.desc = torch.from_numpy(np.concatenate(rows).astype(np.int64)).to(self.lib.device)
        self.gptrs = [g_.data_ptr() for _, g_, _, _ in self.tensors]

    def step(self, lr=None, grads=None):
        """One optimiser step.  grads (optional): this step's gradient dict -- the table was built from the tensors given to the constructor;
        when a tensor of `grads` lives elsewhere (a step that handed out fresh gradient tensors) the table is rebuilt instead of silently
        applying stale or cleared memory."""
        if grads is not None:
            assert set(grads) == set(self.keys), "AdamW: the gradient set changed"
            if any(grads[k].data_ptr() != q for k, q in zip(self.keys, self.gptrs)):
                self._rebind(grads)
        self.t += 1
        lr_, b1, b2, eps, wd = self.hp
        lib = self.lib
        with lib.on_stream():
            lib.check(lib.dll.mugd_train_adamw_chunks(lib.ctx, self.desc.data_ptr(), self.nchunks, float(lr_ if lr is None else lr), b1, b2, eps, wd, self.t))


# ------------------------------------------------------------------------------------------------------------------------------
# whole-model training step (mug/diffusion/diffusion.py:356-414 DDPM.p_losses / training_step)
# ------------------------------------------------------------------------------------------------------------------------------
_SUB_CACHE = {}


def _sub(sd, q):
    """The tensors under module prefix q, keyed relative to it.  Memoised per state dict object (a step asks for every block
    twice; scanning 1515 names each time was a measurable part of a small-batch step).  Tensors must be updated IN PLACE (as
    adamw_step does): an entry replaced in the dict afterwards is not seen."""
    key = id(sd)
    ent = _SUB_CACHE.get(key)
    if ent is None or ent[0] is not sd:
        if len(_SUB_CACHE) > 4:
            _SUB_CACHE.clear()
        ent = _SUB_CACHE[key] = (sd, {})
    hit = ent[1].get(q)
    if hit is not None:
        n = len(q) + 1
        for k, v in hit.items():                    # an entry REPLACED in the dict since (sd[k] = v.to(dev), a reload) must not be served stale
            if sd.get(q + "." + k) is not v:
                hit = None
                break
    if hit is None:
        n = len(q) + 1
        hit = ent[1][q] = {k[n:]: v for k, v in sd.items() if k.startswith(q + ".")}
    return hit


def invalidate_param_cache(sd=None):
    """Forget the memoised per-block parameter views (of `sd`, or all)."""
    if sd is None:
        _SUB_CACHE.clear()
    else:
        _SUB_CACHE.pop(id(sd), None)


class FlatGrads:
    """One zeroed buffer for all parameter gradients of a step; `take(like)` hands out the next slice shaped like a parameter
    (None when the buffer is exhausted: the caller falls back to its own allocation)."""

    def __init__(self, numel, device):
        self.buf = torch.zeros(int(numel), dtype=torch.float32, device=device)
        self.off = 0

    def take(self, like):
        n = like.numel()
        n_al = (n + 63) // 64 * 64                     # 256-byte aligned slices
        if like.dtype != torch.float32 or self.off + n_al > self.buf.numel():
            return None
        g = self.buf[self.off:self.off + n].view(like.shape)
        self.off += n_al
        return g


class _Grads:
    """Gradients keyed by full state-dict name; a tensor reached twice accumulates.

    With `reducer` (a BucketedAllReduce) every tensor is handed over the moment its block's backward has produced it: buckets
    fill in backward order and their all-reduce runs while the sweep continues (every parameter of this model belongs to exactly
    one block, so a gradient is final when it first appears)."""

    def __init__(self, reducer=None):
        self.g = {}
        self.reducer = reducer

    def add(self, prefix, grads):
        for k, v in grads.items():
            key = prefix + "." + k if prefix else k
            if key in self.g:
                assert self.reducer is None, "gradient of %s produced twice: cannot be reduced early" % key
                self.g[key] = self.g[key] + v          # never happens in this model (every parameter belongs to one block)
            else:
                self.g[key] = v
                if self.reducer is not None:
                    self.reducer.push(v)


class BucketedAllReduce:
    """DDP-style gradient reduction overlapped with the backward sweep: gradients are appended to the open bucket as they are
    produced; a full bucket (`bucket_bytes`) is flattened and all-reduced asynchronously (RCCL over xGMI under the `nccl` backend:
    the collective runs on RCCL's own stream next to the backward kernels; `gloo` in the CPU tests); `finish()` flushes the last
    bucket, waits for all of them and writes the averaged values back into the gradient tensors.  Ranks produce gradients in the
    same order (same model, same sweep), so bucket boundaries agree without negotiation."""

    def __init__(self, bucket_bytes=64 << 20, average=True, group=None, even_single=False):
        """even_single: run the collectives in a one-rank group too (a dry run of the RCCL path on a single GPU; a no-op otherwise)."""
        self.bucket_bytes, self.average, self.group = bucket_bytes, average, group
        self.open, self.open_bytes, self.inflight = [], 0, []
        self.on = dist.is_available() and dist.is_initialized() and (dist.get_world_size(group) > 1 or even_single)
        self.n_buckets = 0
        self.before_launch = None            # TrainPlan.step: the library's queued gradient reductions run before a bucket is read

    def push(self, t):
        if not self.on:
            return
        self.open.append(t)
        self.open_bytes += t.numel() * t.element_size()
        if self.open_bytes >= self.bucket_bytes:
            self._launch()

    def _launch(self):
        if not self.open:
            return
        if self.before_launch is not None:
            self.before_launch()
        dev = self.open[0].device
        backend_dev = dev if dist.get_backend(self.group) == "nccl" else torch.device("cpu")
        flat = torch.cat([t.reshape(-1).to(backend_dev) for t in self.open])
        work = dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
        self.inflight.append((work, flat, self.open))
        self.open, self.open_bytes = [], 0
        self.n_buckets += 1

    def finish(self):
        if not self.on:
            return
        self._launch()
        world = dist.get_world_size(self.group)
        for work, flat, tensors in self.inflight:
            work.wait()
            if self.average:
                flat /= world
            off = 0
            for t in tensors:
                n = t.numel()
                t.copy_(flat[off:off + n].reshape(t.shape).to(t.device))
                off += n
        self.inflight = []


def unet_plan(cfg):
    """The U-Net constructor's module order (mug/diffusion/unet.py:341-487), index-compatible with the state-dict keys:
    input_blocks = conv_in, then per level [audio concat][num_res_blocks x (res, [attn], [s4])][down]; output_blocks mirrored with
    one more sequence per level, no S4 in the last one, the upsample appended to it."""
    mc, mult, nrb, attn_res = cfg["model_channels"], cfg["channel_mult"], cfg["num_res_blocks"], cfg["attention_resolutions"]
    inp, ds = [("conv_in",)], 1
    for level in range(len(mult)):
        inp.append(("audio", level))
        for _ in range(nrb):
            inp.append(("seq", ["res"] + (["attn"] if ds in attn_res else []) + (["s4"] if cfg.get("s4_layer", False) else [])))
        if level != len(mult) - 1:
            inp.append(("down",))
            ds *= 2
    out = []
    for level in reversed(range(len(mult))):
        out.append(("audio", level))
        for i in range(nrb + 1):
            layers = ["res"] + (["attn"] if ds in attn_res else [])
            if cfg.get("s4_layer", False) and i != nrb:
                layers.append("s4")
            if level and i == nrb:
                layers.append("up")
                ds //= 2
            out.append(("seq", layers))
    return inp, out


class UNetStep:
    """Forward sweep / backward sweep of UNetModel.forward (unet.py:511-550) over the native block entry points."""

    def __init__(self, lib, sd, cfg, prefix="model.unet_model", groups=32, recompute=False, packs=None):
        self.lib, self.sd, self.cfg, self.p, self.groups, self.recompute = lib, sd, cfg, prefix, groups, recompute
        self.packs = packs          # {block name: argument pack} kept across steps by a TrainPlan; None: rebuilt per call

    def _pk(self, q):
        return None if self.packs is None else self.packs.setdefault(q, {})

    def _st(self):
        """recompute=False: the block keeps its forward intermediates between the two sweeps (a TrainState); True: block-level
        checkpointing, the backward call recomputes the block's forward."""
        from ._native import TrainState
        return None if self.recompute else TrainState()

    def _seq_forward(self, q, layers, h, emb, ctx, tape):
        lib, heads = self.lib, self.cfg["num_heads"]
        for j, kind in enumerate(layers):
            qq = "%s.%d" % (q, j)
            st = self._st()
            tape.append((kind, qq, h, st))
            if kind == "res":
                h = lib.train_resblock(_sub(self.sd, qq), h, emb, None, groups=self.groups, state=st, pack=self._pk(qq))[0]
            elif kind == "attn":
                h = lib.train_transformer(_sub(self.sd, qq), h, ctx, None, heads, groups=self.groups, state=st, pack=self._pk(qq))[0]
            elif kind == "s4":
                h = lib.train_s4layer(_sub(self.sd, qq), h, None, groups=self.groups, state=st, pack=self._pk(qq))[0]
            else:
                h = lib.train_conv(self.sd[qq + ".conv.weight"], self.sd[qq + ".conv.bias"], h, None, mode=2, state=st, pack=self._pk(qq + ".conv"))[0]
        return h

    def forward(self, x, t, context, audios):
        """x (B, C, z), t (B) long, context (B, Cc, ntok), audios: the wave encoder's maps (the last len(channel_mult) are used)."""
        lib, sd, cfg, p = self.lib, self.sd, self.cfg, self.p
        nl = len(cfg["channel_mult"])
        self.temb = lib.op_timestep_embedding(t, cfg["model_channels"])
        self.emb, _ = lib.train_time_embed(_sub(sd, p + ".time_embed"), self.temb, None, pack=self._pk(p + ".time_embed"))
        self.context, self.audios = context, [a.to(lib.device) for a in audios]
        inp, out = unet_plan(cfg)
        tape, hs, h, ai = [], [], lib.f32(x), -nl
        for i, mod in enumerate(inp):
            q = "%s.input_blocks.%d" % (p, i)
            if mod[0] == "audio":
                tape.append(("audio_cat", h.shape[1], ai))
                h = lib.train_concat(h, self.audios[ai])
                ai += 1
                continue
            if mod[0] == "conv_in":
                st = self._st()
                tape.append(("conv", q + ".0", h, 0, st))
                h = lib.train_conv(sd[q + ".0.weight"], sd[q + ".0.bias"], h, None, state=st, pack=self._pk(q + ".0"))[0]
            elif mod[0] == "down":
                st = self._st()
                tape.append(("conv", q + ".0.conv", h, 1, st))
                h = lib.train_conv(sd[q + ".0.conv.weight"], sd[q + ".0.conv.bias"], h, None, mode=1, state=st, pack=self._pk(q + ".0.conv"))[0]
            else:
                h = self._seq_forward(q, mod[1], h, self.emb, context, tape)
            tape.append(("push", len(hs)))
            hs.append(h)
        ai = -1
        q = p + ".middle_block"
        h = self._seq_forward(q, ["res", "attn", "res"], h, self.emb, context, tape)
        for i, mod in enumerate(out):
            q = "%s.output_blocks.%d" % (p, i)
            if mod[0] == "audio":
                tape.append(("audio_cat", h.shape[1], ai))
                h = lib.train_concat(h, self.audios[ai])
                ai -= 1
                continue
            tape.append(("skip_cat", h.shape[1], len(hs) - 1))
            h = lib.train_concat(h, hs.pop())
            h = self._seq_forward(q, mod[1], h, self.emb, context, tape)
        st = self._st()
        tape.append(("out", p + ".out", h, st))
        y = lib.train_conv(sd[p + ".out.2.weight"], sd[p + ".out.2.bias"], h, None, gn=(sd[p + ".out.0.weight"], sd[p + ".out.0.bias"]),
                           groups=self.groups, state=st, pack=self._pk(p + ".out"))[0]
        self.tape = tape
        return y

    def backward(self, dy, grads):
        """dy: gradient of the output.  Fills `grads` (_Grads); returns (dx, dcontext, {audio index: gradient})."""
        lib, sd, heads = self.lib, self.sd, self.cfg["num_heads"]
        dh, demb, dctx, daud, skips = dy, None, None, {}, {}
        for rec in reversed(self.tape):
            kind = rec[0]
            if kind == "out":
                _, q, h, st = rec
                _, dh, dw, db, dg = lib.train_conv(sd[q + ".2.weight"], sd[q + ".2.bias"], h, dh, gn=(sd[q + ".0.weight"], sd[q + ".0.bias"]), groups=self.groups,
                                                   state=st, pack=self._pk(q))
                grads.add(q, {"2.weight": dw, "2.bias": db, "0.weight": dg[0], "0.bias": dg[1]})
            elif kind == "conv" or kind == "up":
                q, h, mode = (rec[1], rec[2], rec[3]) if kind == "conv" else (rec[1] + ".conv", rec[2], 2)
                _, dh, dw, db, _ = lib.train_conv(sd[q + ".weight"], sd[q + ".bias"], h, dh, mode=mode, state=rec[-1], pack=self._pk(q))
                grads.add(q, {"weight": dw, "bias": db})
            elif kind == "res":
                _, dh, de, g = lib.train_resblock(_sub(sd, rec[1]), rec[2], self.emb, dh, groups=self.groups, state=rec[3], pack=self._pk(rec[1]))
                demb = de if demb is None else lib.train_add(demb, de, out=demb)
                grads.add(rec[1], g)
            elif kind == "attn":
                _, dh, dc, g = lib.train_transformer(_sub(sd, rec[1]), rec[2], self.context, dh, heads, groups=self.groups, state=rec[3], pack=self._pk(rec[1]))
                if dc is not None:
                    dctx = dc if dctx is None else lib.train_add(dctx, dc, out=dctx)
                grads.add(rec[1], g)
            elif kind == "s4":
                _, dh, g = lib.train_s4layer(_sub(sd, rec[1]), rec[2], dh, groups=self.groups, state=rec[3], pack=self._pk(rec[1]))
                grads.add(rec[1], g)
            elif kind == "skip_cat":                      # gradient of cat([h, skip]): two channel slices
                dh, skips[rec[2]] = lib.train_split(dh, rec[1])
            elif kind == "audio_cat":                     # gradient of cat([h, audio map]); a map read on the way down and up accumulates
                dh, daud[rec[2]] = lib.train_split(dh, rec[1], acc_b=daud.get(rec[2]))
            elif kind == "push":                          # the tensor went to the next block AND onto the skip stack
                dh = lib.train_add(dh, skips.pop(rec[1]), out=dh)
        _, g = lib.train_time_embed(_sub(sd, self.p + ".time_embed"), self.temb, demb, pack=self._pk(self.p + ".time_embed"))
        grads.add(self.p + ".time_embed", g)
        return dh, dctx, daud


class WaveStep:
    """MelspectrogramScaleEncoder1D.forward (mug/cond/wave.py:398-464) as a forward / backward sweep."""

    def __init__(self, lib, sd, cfg, prefix="model.wave_model", recompute=False, packs=None):
        self.lib, self.sd, self.cfg, self.p, self.recompute = lib, sd, cfg, prefix, recompute
        self.packs = packs

    def _pk(self, q):
        return None if self.packs is None else self.packs.setdefault(q, {})

    def _st(self):
        from ._native import TrainState
        return None if self.recompute else TrainState()

    def forward(self, mel):
        lib, sd, cfg, p = self.lib, self.sd, self.cfg, self.p
        g, heads = cfg["num_groups"], cfg["num_heads"]
        st = self._st()
        tape, hs, ds = [("conv", p + ".conv_in", lib.f32(mel), 0, st)], [], 1
        h = lib.train_conv(sd[p + ".conv_in.weight"], sd[p + ".conv_in.bias"], mel, None, state=st, pack=self._pk(p + ".conv_in"))[0]
        for lvl in range(len(cfg["channel_mult"])):
            q = "%s.down.%d" % (p, lvl)
            if lvl != 0:
                st = self._st()
                tape.append(("conv", q + ".downsample.conv", h, 1, st))
                h = lib.train_conv(sd[q + ".downsample.conv.weight"], sd[q + ".downsample.conv.bias"], h, None, mode=1, state=st, pack=self._pk(q + ".downsample.conv"))[0]
                ds *= 2
            for ib in range(cfg["num_res_blocks"]):
                dil = (1, 2) if ib % 2 == 0 else (4, 8)
                qq = "%s.block.%d" % (q, ib)
                st = self._st()
                tape.append(("resnet", qq, h, dil, st))
                h = lib.train_resnet_block(_sub(sd, qq), h, None, groups=g, dilations=dil, state=st, pack=self._pk(qq))[0]
                if ds in cfg["attention_resolutions"]:
                    qq = "%s.attn.%d" % (q, ib)
                    st = self._st()
                    tape.append(("attn", qq, h, st))
                    h = lib.train_transformer(_sub(sd, qq), h, None, None, heads, groups=32, state=st, pack=self._pk(qq))[0]
            tape.append(("emit", lvl))
            hs.append(h)
        self.tape = tape
        return hs

    def backward(self, dhs, grads):
        """dhs: {level index (negative, as the U-Net addresses them, or non-negative): gradient}."""
        lib, sd, cfg = self.lib, self.sd, self.cfg
        nl = len(cfg["channel_mult"])
        want = {(k % nl): v for k, v in dhs.items()}
        dh = None
        for rec in reversed(self.tape):
            kind = rec[0]
            if kind == "emit":
                if rec[1] in want:
                    dh = want[rec[1]] if dh is None else lib.train_add(dh, want[rec[1]], out=dh)
            elif dh is None:
                continue                                    # levels above the last one the U-Net reads get no gradient
            elif kind == "conv":
                _, dh, dw, db, _ = lib.train_conv(sd[rec[1] + ".weight"], sd[rec[1] + ".bias"], rec[2], dh, mode=rec[3], state=rec[4], pack=self._pk(rec[1]))
                grads.add(rec[1], {"weight": dw, "bias": db})
            elif kind == "resnet":
                _, dh, g = lib.train_resnet_block(_sub(sd, rec[1]), rec[2], dh, groups=cfg["num_groups"], dilations=rec[3], state=rec[4], pack=self._pk(rec[1]))
                grads.add(rec[1], g)
            elif kind == "attn":
                _, dh, _, g = lib.train_transformer(_sub(sd, rec[1]), rec[2], None, dh, cfg["num_heads"], groups=32, state=rec[3], pack=self._pk(rec[1]))
                grads.add(rec[1], g)
        return dh


class TrainPlan:
    """Everything about a training step that does not change from step to step, built once: the two sweep objects with one argument
    pack per block (parameter / gradient pointer arrays, gradient views) and ONE gradient arena all parameter gradients are views of
    (cleared by a single fill per step).  `step()` returns (loss, grads) where grads are views of the arena: they are overwritten by
    the next `step()` -- consume them (AdamW, all-reduce) first.  The parameters of `sd` must be updated IN PLACE (adamw_step does);
    call `invalidate()` after replacing tensors in `sd`."""

    def __init__(self, lib, sd, unet_cfg, wave_cfg, recompute=False):
        self.lib, self.sd, self.unet_cfg, self.wave_cfg, self.recompute = lib, sd, unet_cfg, wave_cfg, recompute
        self.invalidate()

    def invalidate(self):
        """Forget everything derived from the tensors of `sd` (call after replacing / moving parameters).  Also drops the library's
        packed-weight cache, which is keyed by tensor address and re-read by the next bracketed step (include/mugd.h), and makes this
        plan the owner the binding keeps alive while that cache may point into its tensors."""
        sd = self.sd
        invalidate_param_cache(sd)
        self.lib.train_step_reset(owner=self)
        n_train = sum(v.numel() + 64 for k, v in sd.items() if v.dtype == torch.float32 and k.startswith("model.") and not k.startswith("model.first_stage_model."))
        self.arena = FlatGrads(n_train, self.lib.device)
        self.packs = {}
        self.first = True
        self._throttle = None         # event behind the previous step's backward sweep (step(): the host stays less than one step ahead)
        # the library's step bracket keys its packed-weight cache on tensor addresses: only when every parameter already lives on the
        # library's device as contiguous fp32 (no per-call staging copies whose addresses could be recycled inside a step)
        dev = torch.device(self.lib.device)
        self.bracket = all(v.device.type == dev.type and v.dtype == torch.float32 and v.is_contiguous()
                           for k, v in sd.items() if torch.is_tensor(v) and v.is_floating_point() and k.startswith("model.") and not k.startswith("model.first_stage_model."))

    def step(self, x0, noise, t, ids, mel, beta=0.02, add=0.01, reducer=None):
        lib = self.lib
        if lib._bracket_owner is not self:            # another plan (another model's tensors) used the bracket since: its cache entries go first
            lib.train_step_reset(owner=self)
        with lib.on_stream():                         # torch's fills / allocations and the library's kernels on ONE stream: no event per call
            # Nothing inside a step waits for the device (mugd_train_release_states stopped draining the stream in round 6), so a loop of steps
            # would let the host run as far ahead as the runtime's queue allows -- and a flooded queue makes the DEVICE slower (measured at batch
            # 32: 74.5 ms per step against 63 with a host that waits once per step; profiles/r6_train_step_digest*.txt).  The wait that keeps the
            # device fed: before enqueueing step N + 1, until the device has finished step N's BACKWARD sweep -- its reduction-table launch and
            # AdamW are still queued then, so the stream never runs dry while the host catches up.
            if self._throttle is not None:
                self._throttle.synchronize()
                self._throttle = None
            lib.train_release_states()                # intermediates of a sweep that was abandoned half-way
            if not self.first:
                self.arena.buf.zero_()                # packs hold views of the arena: clear it, keep the layout
            self.first = False
            lib.grad_arena = self.arena
            # the step bracket: no parameter changes until step_end, so packed weights come from the library's cache (one refresh launch
            # here) and the split-K / bias-row reductions of the weight gradients run as one launch per flush
            if self.bracket:
                lib.train_step_begin()
                if reducer is not None:
                    reducer.before_launch = lib.train_step_flush
            try:
                return _training_step(lib, self.sd, self.unet_cfg, self.wave_cfg, x0, noise, t, ids, mel, beta, add, reducer, self.recompute,
                                      _Grads(reducer), self.packs, mark=self._mark)
            finally:
                if self.bracket:
                    lib.train_step_end()
                if reducer is not None:
                    reducer.before_launch = None
                lib.grad_arena = None


    def _mark(self):
        """Called by _training_step behind the backward sweep: the point the next step() waits for."""
        if self.lib.device.type == "cuda":
            self._throttle = torch.cuda.Event()
            self._throttle.record(torch.cuda.current_stream(self.lib.device))


def training_step(lib, sd, unet_cfg, wave_cfg, x0, noise, t, ids, mel, beta=0.02, add=0.01, reducer=None, recompute=False):
    """One DDPM training step's loss and gradients (diffusion.py:356-414: x_t = q_sample(x0, t, noise); eps = unet(x_t, t,
    cond(ids), *wave(mel)); loss = mean_b(mean smooth_l1(noise, eps; beta) + add)), for every trainable tensor of the U-Net, the
    wave encoder and the prompt-feature embedding table.  sd: the model's state dict (full names, device or host tensors).
    reducer: a BucketedAllReduce -- the gradients are then all-reduced bucket by bucket WHILE the backward sweep runs and come
    back averaged over the ranks.  recompute: False -- every block keeps its forward intermediates for its backward call (one forward
    + one backward; a few GB at batch 32); True -- block-level activation checkpointing (the backward call recomputes the block's
    forward: only block inputs are stored).  Both give bit-identical gradients.  Returns (loss, {state-dict name: gradient}).
    This functional form builds a throw-away TrainPlan (fresh gradient buffer, nothing cached across calls); loops use TrainPlan.step."""
    return TrainPlan(lib, sd, unet_cfg, wave_cfg, recompute=recompute).step(x0, noise, t, ids, mel, beta=beta, add=add, reducer=reducer)


def _training_step(lib, sd, unet_cfg, wave_cfg, x0, noise, t, ids, mel, beta, add, reducer, recompute, grads, packs=None, mark=None):
    xt = lib.train_q_sample(x0, noise, t, sd["sqrt_alphas_cumprod"], sd["sqrt_one_minus_alphas_cumprod"])
    table = lib.f32(sd["model.cond_stage_model.embedding.weight"])
    context = lib.cond_embed(table, ids)
    wave = WaveStep(lib, sd, wave_cfg, recompute=recompute, packs=packs)
    audios = wave.forward(mel)
    unet = UNetStep(lib, sd, unet_cfg, recompute=recompute, packs=packs)
    pred = unet.forward(xt, t, context, audios)
    loss, dpred = lib.train_smooth_l1(pred, noise, beta=beta, add=add)
    _, dctx, daud = unet.backward(dpred, grads)
    grads.add("", {"model.cond_stage_model.embedding.weight": lib.train_embedding_bwd(ids, dctx, table.shape[0],
                                                                                     pack=None if packs is None else packs.setdefault("#embedding", {}))})
    wave.backward(daud, grads)
    if mark is not None:
        mark()
    if reducer is not None:
        reducer.finish()
    lib.train_release_states()                       # wave-encoder levels the U-Net does not read never run their backward
    return loss.mean(), grads.g


_VAE_ENC = {}


def encode_x0(lib, sd, vae_cfg, note):
    """The frozen first stage in front of the step, as DDPM.forward does (diffusion.py:408: `x = self.model.encode(batch)`, then
    p_losses :357 `x_start = x_start_distribution.mode()`): AutoencoderKL.encode (autoencoder.py:67-73) on the native VaeEncoder
    network; the mode of the diagonal Gaussian is its mean = the first z_channels rows of the moments.  The encoder instance is
    kept per (library, state dict)."""
    key = (id(lib), id(sd))
    ent = _VAE_ENC.get(key)
    if ent is None or ent[0] is not sd:
        if len(_VAE_ENC) > 2:
            _VAE_ENC.clear()
        enc = lib.vae(vae_cfg, encoder=True)
        enc.set_params(sd, "model.first_stage_model.")
        ent = _VAE_ENC[key] = (sd, enc)
    moments = ent[1].vae_encode(note)
    return moments[:, :vae_cfg["z_channels"]].contiguous()


def training_step_from_batch(lib, sd, unet_cfg, wave_cfg, vae_cfg, batch, t, noise, **kw):
    """DDPM.forward + p_losses on a reference-shaped batch {'note': (B, x_ch, 8 z), 'audio': (B, n_freq, frames), 'feature': (B, 21) ids}
    (diffusion.py:356-414) with explicit (t, noise) -- what the reference draws from its generator: the frozen VAE encoder's
    mode() in front, then `training_step`."""
    x0 = encode_x0(lib, sd, vae_cfg, batch["note"])
    return training_step(lib, sd, unet_cfg, wave_cfg, x0, noise, t, batch["feature"].long(), batch["audio"], **kw)


# ------------------------------------------------------------------------------------------------------------------------------
# a minimal training loop over the step (the reference trains through pytorch_lightning: main.py + DDPM.training_step /
# configure_optimizers, diffusion.py:416-513; its dataset is not available offline -- SURVEY.md 8f rank 4 -- so: synthetic batches)
# ------------------------------------------------------------------------------------------------------------------------------
def synthetic_batch(B, z, unet_cfg, wave_cfg, n_ids, ntok, audio_ratio, seed, device):
    """One synthetic training batch with the shapes of the real one: x0 (B, z_channels, z) latent, mel (B, n_freq, z * audio_ratio)
    non-negative log-mel, ids (B, ntok) prompt-feature ids."""
    g = torch.Generator().manual_seed(seed)
    x0 = torch.randn(B, unet_cfg["in_channels"], z, generator=g)
    mel = torch.randn(B, wave_cfg["n_freq"], z * audio_ratio, generator=g).abs()
    ids = torch.randint(0, n_ids, (B, ntok), generator=g)
    return x0.to(device), mel.to(device), ids.to(device)


def fit(lib, sd, unet_cfg, wave_cfg, steps, batch, z, lr=1e-4, weight_decay=0.01, seed=0, fixed_batch=False, audio_ratio=None, ntok=21,
        log=None, recompute=False):
    """`steps` optimiser steps of DDPM training (diffusion.py:404-414: t ~ U{0..T-1}, noise ~ N(0, I), p_losses; AdamW(lr) like
    configure_optimizers :477-499) on synthetic batches, data parallel when torch.distributed is initialised (per-rank batches,
    gradients averaged by BucketedAllReduce during the backward sweep).  Updates the tensors of `sd` in place.  Returns the per-step
    losses of this rank."""
    rank = dist.get_rank() if dist.is_available() and dist.is_initialized() else 0
    world = dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1
    nl = len(unet_cfg["channel_mult"])
    if audio_ratio is None:                  # mel frames per latent sample: the wave encoder's level that feeds U-Net level 0
        audio_ratio = 2 ** (len(wave_cfg["channel_mult"]) - nl)
    n_ids = sd["model.cond_stage_model.embedding.weight"].shape[0]
    T = sd["sqrt_alphas_cumprod"].shape[0]
    dev = lib.device
    for k, v in list(sd.items()):
        if v.dtype == torch.float32 and k.startswith("model.") and v.device != dev:
            sd[k] = v.to(dev)
    invalidate_param_cache(sd)
    broadcast_parameters(sd)                         # ranks start from rank 0's weights, as under DistributedDataParallel
    plan, opt, losses = TrainPlan(lib, sd, unet_cfg, wave_cfg, recompute=recompute), None, []
    g = torch.Generator().manual_seed(seed * 1000 + rank)
    for i in range(steps):
        bseed = seed * 7919 + (0 if fixed_batch else i) * world + rank
        x0, mel, ids = synthetic_batch(batch, z, unet_cfg, wave_cfg, n_ids, ntok, audio_ratio, bseed, dev)
        if fixed_batch:                                   # the same (t, noise) too: a pure overfitting run
            g = torch.Generator().manual_seed(seed * 1000 + rank)
        t = torch.randint(0, T, (batch,), generator=g)
        noise = torch.randn(batch, unet_cfg["in_channels"], z, generator=g).to(dev)
        red = BucketedAllReduce() if world > 1 else None
        loss, grads = plan.step(x0, noise, t, ids, mel, reducer=red)
        if opt is None:                                   # the plan's gradient tensors are the same views every step (AdamW.step checks)
            opt = AdamW(lib, {k: sd[k] for k in grads}, grads, lr=lr, weight_decay=weight_decay)
        opt.step(grads=grads)
        losses.append(float(loss))
        if log is not None and rank == 0:
            log(i, losses[-1])
    return losses


def main(argv=None):
    """python -m mug.train --config model.yaml [--ckpt model.ckpt | --synthetic-seed 0] --steps 10 --batch 32
    (under `python -m torch.distributed.run --nproc-per-node N ...`: data parallel over N GPUs, RCCL)."""
    import argparse
    import os
    import time
    from . import job
    from ._native import get_lib
    ap = argparse.ArgumentParser(prog="python -m mug.train", description=main.__doc__)
    ap.add_argument("--config", required=True)
    ap.add_argument("--ckpt", default=None)
    ap.add_argument("--synthetic-seed", type=int, default=None)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--batch", type=int, default=32, help="per-GPU batch")
    ap.add_argument("--z", type=int, default=None, help="latent length (default: the config's z_length)")
    ap.add_argument("--lr", type=float, default=1e-4)
    ap.add_argument("--recompute", action="store_true", help="block-level activation checkpointing")
    ap.add_argument("--save", default=None, help="write the trained state dict here (rank 0)")
    ap.add_argument("--fp32", action="store_true", help="fp32-input MFMA GEMMs (the parity mode) instead of bf16 (BASELINE configs[4])")
    a = ap.parse_args(argv)
    if a.ckpt is None and a.synthetic_seed is None:
        ap.error("give --ckpt or --synthetic-seed: every rank must build the same initial weights")
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    lib = get_lib()
    lib.train_set_precision(not a.fp32)
    model, cfg = job.load_model(a.config, a.ckpt, device="cuda", seed_synthetic=a.synthetic_seed if a.ckpt is None else None)
    mp = (cfg["model"] if "model" in cfg else cfg)["params"]
    sd = {k: v.detach() for k, v in model.state_dict().items()}
    z = a.z or int(mp.get("z_length", 512))
    t0 = time.perf_counter()

    def log(i, loss):
        print("step %4d  loss %.6f  %.1f samples/s" % (i, loss, (i + 1) * a.batch * world / (time.perf_counter() - t0)), flush=True)

    fit(lib, sd, mp["unet_config"]["params"], mp["wave_stage_config"]["params"], a.steps, a.batch, z, lr=a.lr, seed=a.synthetic_seed or 0,
        log=log, recompute=a.recompute)
    if a.save and (not dist.is_initialized() or dist.get_rank() == 0):
        torch.save({"state_dict": {k: v.cpu() for k, v in sd.items()}}, a.save)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
