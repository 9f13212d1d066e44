"""Batch sharding of one solve across the GPUs of a node (one process per GPU, RCCL over xGMI).

Trajectories are independent (the reference never mixes batch rows: euler.py:36, milstein.py:72, srk.py:74-87,
midpoint.py:39-43), so rank r simply solves rows [r*B/N, (r+1)*B/N). The counter RNG indexes its normals by the
GLOBAL row, so the gathered result is bit-identical to an unsharded solve for any N. The only communication
is one ``all_gather_into_tensor`` of the requested outputs after the last step (for the adjoint: one
``all_reduce`` of the parameter gradients, which autograd users get from DDP-style hooks or by calling
``all_reduce_gradients`` below). Nothing is exchanged per step.
"""
import torch
import torch.distributed as dist

from . import contract
from .brownian import BrownianInterval
from .integrate import sdeint
from .settings import NOISE_TYPES


def shard_rows(global_batch, world_size, rank):
    """Contiguous, balanced row range [r0, r1) of `rank` (the first `global_batch % world_size` ranks get one more)."""
    base, rem = divmod(global_batch, world_size)
    r0 = rank * base + min(rank, rem)
    return r0, r0 + base + (1 if rank < rem else 0)


def _noise_channels(sde, y0, ts):
    if sde.noise_type == NOISE_TYPES.diagonal:
        return y0.size(1)
    if sde.noise_type == NOISE_TYPES.scalar:
        return 1
    with torch.no_grad():
        return sde.g(ts[0] if torch.is_tensor(ts) else torch.tensor(ts[0], dtype=y0.dtype, device=y0.device),
                     y0[:1]).size(-1)


def gather_rows(local, global_batch, group=None):
    """All-gather row blocks produced by `shard_rows` into one (global_batch, ...) tensor on every rank."""
    world = dist.get_world_size(group)
    if world == 1:
        return local
    rank = dist.get_rank(group)
    sizes = [shard_rows(global_batch, world, r) for r in range(world)]
    counts = [b - a for a, b in sizes]
    local = local.contiguous()
    if len(set(counts)) == 1:
        out = local.new_empty((global_batch,) + tuple(local.shape[1:]))
        dist.all_gather_into_tensor(out, local, group=group)
        return out
    # ragged shards: pad to the largest block, gather, trim
    width = max(counts)
    padded = local.new_zeros((width,) + tuple(local.shape[1:]))
    padded[:counts[rank]] = local
    out = local.new_empty((world * width,) + tuple(local.shape[1:]))
    dist.all_gather_into_tensor(out, padded, group=group)
    return torch.cat([out[r * width:r * width + counts[r]] for r in range(world)], dim=0)


def sdeint_sharded(sde, y0_global, ts, *, entropy, method=None, dt=1e-3, group=None, gather="final",
                   levy_area_approximation=None, solve_fn=None, **kwargs):
    """Solve this rank's rows of `y0_global` and gather.

    Args:
        y0_global: the full (B, d) initial state (every rank passes the same tensor; only its rows are read).
        entropy: Brownian seed; MUST be the same on all ranks.
        gather: "final" -> (B, d) final states on every rank; "all" -> (T, B, d); None -> local ys only.
        solve_fn: test hook; defaults to ``torchsde_amd.sdeint``.
    Remaining kwargs go to ``sdeint``.
    """
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    B = y0_global.size(0)
    r0, r1 = shard_rows(B, world, rank)
    y0 = y0_global[r0:r1].contiguous()
    ts_t = ts if torch.is_tensor(ts) else torch.tensor(ts, dtype=y0.dtype, device=y0.device)
    if levy_area_approximation is None:      # what `sdeint` itself would pick for bm=None
        levy_area_approximation = contract.default_levy_area_approximation(
            method if method is not None else contract.default_method(sde))
    m = _noise_channels(sde, y0, ts_t)
    if kwargs.get("logqp") and sde.noise_type == NOISE_TYPES.diagonal:
        m += 1          # the KL column is a state channel, and diagonal noise has one Brownian channel per state channel
    bm = BrownianInterval(t0=ts_t[0], t1=ts_t[-1], size=(r1 - r0, m), dtype=y0.dtype, device=y0.device,
                          entropy=entropy, levy_area_approximation=levy_area_approximation, row_offset=r0)
    solve = sdeint if solve_fn is None else solve_fn
    ys = solve(sde, y0, ts_t, bm=bm, method=method, dt=dt, **kwargs)
    if isinstance(ys, tuple):        # logqp=True / extra=True: gather the trajectory, hand the rest back as it is
        return (_gather_ys(ys[0], B, world, gather, group),) + tuple(ys[1:])
    return _gather_ys(ys, B, world, gather, group)


def _gather_ys(ys, B, world, gather, group):
    if gather is None or world == 1:
        return ys if gather != "final" else ys[-1]
    if gather == "final":
        return gather_rows(ys[-1], B, group)
    return gather_rows(ys.transpose(0, 1).contiguous(), B, group).transpose(0, 1).contiguous()


def all_reduce_gradients(params, group=None):
    """Sum parameter gradients over ranks after a sharded ``sdeint_adjoint`` backward (a_theta is a sum over rows).

    Every rank reduces over the SAME list: all parameters that require a gradient, a missing `.grad` counting as zeros
    (a rank whose rows never touched a parameter must still take part, or the collective would mismatch), one flat
    buffer per dtype (no silent promotion of mixed-precision parameters). A parameter that NO rank touched keeps
    `.grad = None`, as in an unsharded run (an optimiser skips it instead of applying weight decay / Adam moments to
    it): one more small all-reduce carries, per parameter, whether any rank had a gradient."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return
    params = [p for p in params if p.requires_grad]
    by_dtype = {}
    for p in params:
        by_dtype.setdefault(p.dtype, []).append(p)
    if not params:
        return
    touched = torch.tensor([0.0 if p.grad is None else 1.0 for p in params], device=params[0].device)
    dist.all_reduce(touched, group=group)
    touched = dict(zip(map(id, params), touched.tolist()))
    for dtype in sorted(by_dtype, key=str):
        group_params = by_dtype[dtype]
        flat = torch.cat([(torch.zeros_like(p) if p.grad is None else p.grad).reshape(-1) for p in group_params])
        dist.all_reduce(flat, group=group)
        offset = 0
        for p in group_params:
            chunk = flat[offset:offset + p.numel()].view_as(p)
            if p.grad is not None:
                p.grad.copy_(chunk)
            elif touched[id(p)] > 0:
                p.grad = chunk.clone()
            offset += p.numel()
