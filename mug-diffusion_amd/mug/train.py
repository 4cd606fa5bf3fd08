"""Training slice (BASELINE configs[4]) -- FIRST STEP.  What exists natively (include/mugd.h, "training
slice"): q_sample, the smooth-L1 noise-prediction loss with its gradient, forward + backward of TimestepResBlock and of
ContextualTransformer (`Lib.train_resblock`, `Lib.train_transformer`), an AdamW step; here: the glue that turns them into a
data-parallel step on a block and the gradient all-reduce every rank of a DDP job runs (one flat bucket, averaged).  The backward
of the S4 and resampling layers -- and with them `DDPM.training_step` (mug/diffusion/diffusion.py:356-414) for the whole U-Net --
are not built yet.
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


def adamw_step(lib, params, grads, state, step, lr=1e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.01):
    """torch.optim.AdamW semantics on the block's parameter dict, in place on device tensors; `state` holds exp_avg / exp_avg_sq."""
    for k, p in params.items():
        if k not in state:
            state[k] = (torch.zeros_like(p), torch.zeros_like(p))
        m, v = state[k]
        lib.train_adamw(p, grads[k], m, v, step, lr=lr, betas=betas, eps=eps, weight_decay=weight_decay)
    return params
