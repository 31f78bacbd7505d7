"""Candidate-batch sharding across the GPUs of one node.

Candidates are independent units: rank r owns the contiguous block [r*B/N, (r+1)*B/N), the per-primitive
constants are replicated, and the only exchange is one all-gather of the per-rank scores (RCCL over xGMI
on GPUs, gloo in the CPU tests) followed by the reference's first-minimum argmin over the GLOBAL index order
(reference motion_primitive_generator.py:251-257).  Two carriers of the same exchange: torch.distributed (the
*_scores / *_minloc / sharded_best_candidate functions; plumbing only) and, for hosts without torch, the library's
own RCCL entry points (mg_all_gather_scores / mg_sharded_best_candidate over a _capi.Context after dist_init).
The scores come from the scorer callable (libmg_hip on a GPU box).
"""
import numpy as np


def shard_range(n_total, rank, world_size):
    """Contiguous block split; the first (n_total % world_size) ranks get one extra row."""
    base, rem = divmod(int(n_total), int(world_size))
    begin = rank * base + min(rank, rem)
    return begin, begin + base + (1 if rank < rem else 0)


def first_min_argmin(values):
    """First strict minimum; NaN never wins; (0, inf) when nothing wins."""
    v = np.asarray(values, dtype=np.float64)
    if v.size == 0:
        return 0, float("inf")
    w = np.where(np.isnan(v), np.inf, v)
    i = int(np.argmin(w))          # np.argmin returns the first occurrence
    if not np.isfinite(w[i]) and w[i] > 0:
        return 0, float("inf")
    return i, float(w[i])


def all_gather_scores(local_scores, n_total, group=None):
    """All-gather variable-length per-rank score blocks into the global (n_total,) order."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    t = local_scores if isinstance(local_scores, torch.Tensor) else torch.as_tensor(np.asarray(local_scores))
    sizes = [shard_range(n_total, r, world)[1] - shard_range(n_total, r, world)[0] for r in range(world)]
    assert t.numel() == sizes[rank], "local block has %d scores, expected %d" % (t.numel(), sizes[rank])
    m = max(sizes) if sizes else 0
    padded = torch.full((m,), float("inf"), dtype=t.dtype, device=t.device)
    padded[: t.numel()] = t
    gathered = torch.empty((world * m,), dtype=t.dtype, device=t.device)
    dist.all_gather_into_tensor(gathered, padded, group=group)
    parts = [gathered[r * m: r * m + sizes[r]] for r in range(world)]
    return torch.cat(parts) if parts else gathered


def all_gather_minloc(local_index, local_value, offset, group=None):
    """The light exchange: every rank contributes its own first minimum as (global index, value), 16 bytes, and
    every rank picks the smallest value, ties to the smaller global index -- the same winner as the first-minimum
    argmin over the gathered score vector.  A rank with nothing to offer sends (inf, its offset)."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    mine = torch.tensor([float(local_value), float(offset + local_index)], dtype=torch.float64)
    if dist.get_backend(group) == "nccl":
        mine = mine.cuda()
    everyone = torch.empty((world * 2,), dtype=torch.float64, device=mine.device)
    dist.all_gather_into_tensor(everyone, mine, group=group)
    pairs = everyone.cpu().numpy().reshape(world, 2)
    best_v, best_i = float("inf"), None
    for v, i in pairs:                      # ranks in order = global index order: strict '<' keeps the first
        if v < best_v:
            best_v, best_i = float(v), int(i)
    return (0, float("inf")) if best_i is None else (best_i, best_v)


def sharded_best_candidate(samples, scorer, group=None, exchange="scores"):
    """Every rank holds the same `samples` (n, L); each scores its block with `scorer(block) -> (len(block),)`.
    exchange="scores": the scores are all-gathered and every rank returns the same (best_index, min_error,
    all_scores); exchange="minloc": only each rank's (index, value) pair travels and all_scores is None."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    n = len(samples)
    b, e = shard_range(n, rank, world)
    local = scorer(samples[b:e])
    if exchange == "minloc":
        li, lv = first_min_argmin(local.detach().cpu().numpy() if isinstance(local, torch.Tensor) else local)
        idx, val = all_gather_minloc(li, lv, b, group)
        return idx, val, None
    if not isinstance(local, torch.Tensor):
        local = torch.as_tensor(np.asarray(local, dtype=np.float64))
    scores = all_gather_scores(local, n, group)
    idx, val = first_min_argmin(scores.detach().cpu().numpy())
    return idx, val, scores


def mg_all_gather_scores(ctx, local_scores, n_total, rank, world, dtype=np.float64):
    """The all-gather of per-rank score blocks through mg_dist_all_gather (RCCL loaded by libmg_hip.so, communicator
    made by ctx.dist_init): no torch anywhere.  Blocks of different lengths are padded with +inf to the longest."""
    local = np.ascontiguousarray(local_scores, dtype=dtype)
    sizes = [shard_range(n_total, r, world)[1] - shard_range(n_total, r, world)[0] for r in range(world)]
    assert local.size == sizes[rank], "local block has %d scores, expected %d" % (local.size, sizes[rank])
    m = max(sizes) if sizes else 0
    if m == 0:
        return np.empty((0,), dtype=dtype)
    padded = np.full((m,), np.inf, dtype=dtype)
    padded[: local.size] = local
    d_local = ctx.upload(padded)
    d_all = ctx.malloc(world * m * padded.itemsize)
    try:
        ctx.dist_all_gather(d_local, d_all, m, dtype)
        everyone = ctx.download(d_all, (world, m), dtype)     # synchronises the context's stream
    finally:
        d_local.free()
        d_all.free()
    return np.concatenate([everyone[r, : sizes[r]] for r in range(world)])


def mg_sharded_best_candidate(ctx, samples, scorer, rank, world):
    """sharded_best_candidate without torch: every rank scores its contiguous block with `scorer(block)` and the
    blocks travel through mg_all_gather_scores; returns the same (best_index, min_error, all_scores) on every rank."""
    n = len(samples)
    b, e = shard_range(n, rank, world)
    local = np.asarray(scorer(samples[b:e]), dtype=np.float64)
    scores = mg_all_gather_scores(ctx, local, n, rank, world)
    idx, val = first_min_argmin(scores)
    return idx, val, scores
