"""torch.distributed carriers of the sharded argmin's exchange -- TEST HELPERS (tests/test_distributed_gloo.py rehearses the N > 1
exchange on CPU with the gloo backend).  The product package carries no torch: its carriers are the library's RCCL entry points
(distributed.MgCommunicator, mg_all_gather_scores) and files (FileCommunicator)."""
import numpy as np

from morphablegraphs_amd.distributed import shard_range, first_min_argmin


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


