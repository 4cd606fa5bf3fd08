"""mug.diffusion.ddim -- DDIMSampler with the reference's interface (mug/diffusion/ddim.py:11-196).

`sample()` keeps the reference's arguments and return value; the loop itself (ddim_sampling +
p_sample_ddim, :110-196) runs on the device inside libmugd (`mugd_ddim_sample`: the
U-Net program + fused CFG/DDIM update per step, launched eagerly or replayed from a hipGraph), in as few native calls as the arguments allow:

* no callbacks / mask, S <= log_every_t:  ONE call (the library also returns the state after the first step, which the
                         reference logs as an intermediate, :154-156);
* no callbacks / mask, S > log_every_t:   one call for the first step (the reference logs its x / pred_x0 as an
                         intermediate, :154-156), then one call per stretch between logging points;
* callback / img_callback / mask / x0:  one native call per step, with the reference's host-side
                         logic (q_sample blend, callbacks) in between.
"""
import numpy as np
import torch
from tqdm import tqdm

from mug.diffusion.utils import make_ddim_sampling_parameters, make_ddim_timesteps


class DDIMSampler(object):
    def __init__(self, model, schedule="linear", **kwargs):
        # webui.py:105 passes a device as the second positional argument; it lands in `schedule` and is ignored
        super().__init__()
        self.model = model
        self.ddpm_num_timesteps = model.num_timesteps
        self.schedule = schedule
        self.device = model.device

    def register_buffer(self, name, attr):
        if type(attr) == torch.Tensor:
            attr = attr.to(self.model.device)
        setattr(self, name, attr)

    def make_schedule(self, ddim_num_steps, ddim_discretize="uniform", ddim_eta=0., verbose=True):
        """ddim.py:24-53 (only the buffers the sampler reads)."""
        ac = self.model.alphas_cumprod
        key = (ddim_discretize, int(ddim_num_steps), float(ddim_eta), ac.data_ptr(), ac._version)
        if getattr(self, "_sched_key", None) == key:       # same schedule as the last call: no device round trip
            return
        self.ddim_timesteps = make_ddim_timesteps(ddim_discretize, ddim_num_steps, self.ddpm_num_timesteps, verbose=verbose)
        assert ac.shape[0] == self.ddpm_num_timesteps, 'alphas have to be defined for each timestep'
        sig, a, ap = make_ddim_sampling_parameters(ac.detach().cpu().numpy(), self.ddim_timesteps, ddim_eta, verbose=verbose)
        self.ddim_sigmas = np.asarray(sig, dtype=np.float64)
        self.ddim_alphas = np.asarray(a)
        self.ddim_alphas_prev = np.asarray(ap)
        self.ddim_sqrt_one_minus_alphas = np.sqrt(1. - self.ddim_alphas)
        self.register_buffer('betas', self.model.betas.clone().detach().float())
        self.register_buffer('alphas_cumprod', ac.clone().detach().float())
        self.register_buffer('alphas_cumprod_prev', self.model.alphas_cumprod_prev.clone().detach().float())
        self._sched_key = key

    @torch.no_grad()
    def sample(self, S, c, w, batch_size, shape=None, callback=None, img_callback=None, eta=0., mask=None, x0=None,
               temperature=1., noise_dropout=0., verbose=True, x_T=None, log_every_t=100,
               unconditional_guidance_scale=1., unconditional_conditioning=None, tqdm_class=None, **kwargs):
        if c is not None and not isinstance(c, dict) and c.shape[0] != batch_size:
            print(f"Warning: Got {c.shape[0]} conditionings but batch-size is {batch_size}")
        self.make_schedule(ddim_num_steps=S, ddim_eta=eta, verbose=verbose)
        size = (batch_size, self.model.z_channels, self.model.z_length) if shape is None else (batch_size, shape[0], shape[1])
        if verbose:
            print(f'Data shape for DDIM sampling is {size}, eta {eta}')
        return self.ddim_sampling(w, c, size, callback=callback, img_callback=img_callback, mask=mask, x0=x0,
                                  noise_dropout=noise_dropout, temperature=temperature, x_T=x_T, log_every_t=log_every_t,
                                  unconditional_guidance_scale=unconditional_guidance_scale,
                                  unconditional_conditioning=unconditional_conditioning, tqdm_class=tqdm_class)

    def _rows(self):
        """Per loop iteration (time running high -> low): timestep and the fp32 scalars of ddim.py:183-186."""
        ts = self.ddim_timesteps
        n = ts.shape[0]
        rows, steps = [], []
        for i, step in enumerate(np.flip(ts)):
            idx = n - i - 1
            steps.append(int(step))
            rows.append([np.float32(self.ddim_alphas[idx]), np.float32(self.ddim_alphas_prev[idx]),
                         np.float32(self.ddim_sigmas[idx]), np.float32(self.ddim_sqrt_one_minus_alphas[idx])])
        return steps, rows

    @torch.no_grad()
    def ddim_sampling(self, w, c, shape, x_T=None, callback=None, mask=None, x0=None, img_callback=None,
                      log_every_t=100, temperature=1., noise_dropout=0., unconditional_guidance_scale=1.,
                      unconditional_conditioning=None, tqdm_class=None, **kwargs):
        device = self.model.betas.device
        unet = self.model.model.unet_model
        b = shape[0]
        x = torch.randn(shape, device=device) if x_T is None else x_T.to(device)
        unet.prepare_length(shape[-1])
        net = unet.native()
        steps, rows = self._rows()
        total = len(steps)
        uc = unconditional_conditioning if (unconditional_conditioning is not None and unconditional_guidance_scale != 1.) else None
        eta_on = any(float(r[2]) != 0.0 for r in rows)
        per_step = (callback is not None) or (img_callback is not None) or (mask is not None)

        intermediates = {'x_inter': [x], 'pred_x0': [x]}
        # iterations after which the reference records an intermediate (ddim.py:154-156)
        log_after = [i for i in range(total) if (total - i - 1) % log_every_t == 0 or (total - i - 1) == total - 1]
        cuts = sorted(set(range(1, total + 1)) if per_step else set(i + 1 for i in log_after) | {total})
        # the reference's call shape (ddim.py:133-135): an iterable first, then desc / total -- gradio.Progress().tqdm, which
        # webui.py:388 passes in, requires the iterable and has no usable update() / close(); the bar is advanced by pulling
        # from its iterator, n items per native call
        bar_it = iter((tqdm_class or tqdm)(range(total), desc='Charting, using DDIM Sampler', total=total))

        def advance(n):
            for _ in range(n):
                next(bar_it, None)

        def draw_noise(n):
            # one randn(shape) per step, in step order, exactly as p_sample_ddim does (ddim.py:192-194: randn, then the dropout
            # draw when noise_dropout > 0) -- also when eta == 0, where the reference still draws (and discards: sigma = 0) the
            # noise, so that a later sample in the same process starts from the same generator state as in the reference
            # (scripts/mapping.py's n_samples loop).  With eta == 0 nothing is kept: the draws only advance the generator.
            keep = []
            for _ in range(n):
                z = torch.randn(tuple(shape), device=device)
                if noise_dropout > 0.:
                    z = torch.nn.functional.dropout(z * temperature, p=noise_dropout)
                elif eta_on:
                    z = z * temperature
                if eta_on:
                    keep.append(z)
            return torch.stack(keep) if eta_on else None
        if not per_step and total > 1 and log_after == [0, total - 1]:
            # the common case (S <= log_every_t, no callbacks): ONE native call for the whole loop; the library hands back the
            # state after the first step, the only other intermediate the reference records
            noise = draw_noise(total) if eta_on else None
            x, pred, first = net.ddim_sample(x, c, w, steps, rows, uc=uc, scale=float(unconditional_guidance_scale), noise=noise,
                                             want_pred_x0=True, want_first=True)
            if not eta_on:
                draw_noise(total)          # generator bookkeeping only: issued AFTER the loop is enqueued, hidden behind it
            advance(total)
            intermediates['x_inter'] += [first[0], x]
            intermediates['pred_x0'] += [first[1], pred]
            cuts = []
        start = 0
        for end in cuts:
            n = end - start
            if mask is not None:
                assert x0 is not None
                ts = torch.full((b,), steps[start], device=device, dtype=torch.long)
                x = self.model.q_sample(x0, ts) * mask + (1. - mask) * x
            noise = draw_noise(n)
            x, pred = net.ddim_sample(x, c, w, steps[start:end], rows[start:end], uc=uc,
                                      scale=float(unconditional_guidance_scale), noise=noise, want_pred_x0=True)
            advance(n)
            for i in range(start, end):
                if callback:
                    callback(i)
                if img_callback:
                    img_callback(pred, i)
            if (end - 1) in log_after:
                intermediates['x_inter'].append(x)
                intermediates['pred_x0'].append(pred)
            start = end
        advance(1)          # exhausts the iterator: tqdm closes its bar on StopIteration
        return x, intermediates
