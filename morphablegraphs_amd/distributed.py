"""Candidate-batch sharding across the GPUs of one node.

Candidates are independent units: rank r owns the contiguous block [r*B/N, (r+1)*B/N), the per-primitive
constants are replicated, and the only exchange is one all-gather of the per-rank scores (RCCL over xGMI
on GPUs, gloo in the CPU tests) followed by the reference's first-minimum argmin over the GLOBAL index order
(reference motion_primitive_generator.py:251-257).  Two carriers of the same exchange: torch.distributed (the
*_scores / *_minloc / sharded_best_candidate functions; plumbing only) and, for hosts without torch, the library's
own RCCL entry points (mg_all_gather_scores / mg_sharded_best_candidate over a _capi.Context after dist_init).
The scores come from the scorer callable (libmg_hip on a GPU box).
"""
import os
import pickle
import tempfile
import time

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


# ---------------------------------------------------------------------------------------------------------------------
# The product seam on several GPUs ("rank 0 drives; other ranks are workers").
#
# The reference's only process model is one process per core, each with its own graph (examples/
# mg_rest_interface_parallel.py:252-254); the loop that shards is the candidate loop of
# MotionPrimitiveGenerator.evaluate_samples_using_constraints (motion_primitive_generator.py:230-261) and the option loop of
# GraphWalkPlanner (graph_walk_planner.py:184-226).  Here: one process per GPU, every rank holds the same primitives, rank 0
# runs the reference's control flow and, for each scoring step, broadcasts a small command (constraint values, seed, component
# counts -- or the candidates themselves when they come from the host's sklearn stream); every rank scores its contiguous
# block of the candidates; ONE all-gather carries each rank's first minimum {error, global index, winning latent}; every rank
# combines them by (smaller error, then smaller index), which is the first-minimum rule over the global order.
#
# A communicator is two calls -- broadcast_bytes(payload, root) and all_gather_rows(float64 vector) -- so that the same
# algorithm runs over RCCL (MgCommunicator: mg_dist_broadcast / mg_dist_all_gather on the context's stream), over files
# (FileCommunicator: CPU rehearsals and tests, no torch, no GPU) and in one process (LocalCommunicator).
# ---------------------------------------------------------------------------------------------------------------------
class FileRendezvous(object):
    """Bytes between the ranks of one node without torch: files under a common base name, written with an atomic rename.
    Carries the 128-byte RCCL unique id to the ranks (bench.py, worker processes) and, in CPU rehearsals, everything else."""

    def __init__(self, rank, world, base=None, timeout=300.0):
        self.rank, self.world, self.timeout = int(rank), int(world), float(timeout)
        self.base = base or os.path.join(tempfile.gettempdir(), "mg_rdv_%d_%s" % (os.getppid(), os.environ.get("MASTER_PORT", "0")))
        self.seq = 0

    def _path(self, tag, r):
        return "%s.%s.%d" % (self.base, tag, r)

    def put(self, tag, payload):
        tmp = self._path(tag, self.rank) + ".tmp"
        with open(tmp, "wb") as f:
            f.write(payload)
        os.rename(tmp, self._path(tag, self.rank))

    def get(self, tag, r, timeout=None):
        deadline = time.time() + (self.timeout if timeout is None else timeout)
        p = self._path(tag, r)
        while not os.path.exists(p):
            if time.time() > deadline:
                raise TimeoutError("rank %d: timed out waiting for rank %d (%s)" % (self.rank, r, tag))
            time.sleep(0.0005)
        with open(p, "rb") as f:
            return f.read()

    def all_gather(self, payload):
        """every rank's bytes on every rank"""
        tag = "ag%d" % self.seq
        self.seq += 1
        self.put(tag, payload)
        got = [self.get(tag, r) for r in range(self.world)]
        self.put(tag + "done", b"1")
        if self.rank == 0:   # the files go once everybody has read them
            for r in range(self.world):
                self.get(tag + "done", r)
            for r in range(self.world):
                for t in (tag, tag + "done"):
                    try:
                        os.remove(self._path(t, r))
                    except OSError:
                        pass
        return got

    def cleanup(self):
        """remove whatever this rank left behind (a run that ended early)"""
        import glob
        for f in glob.glob(self.base + ".*.%d" % self.rank) + glob.glob(self.base + ".*.%d.tmp" % self.rank):
            try:
                os.remove(f)
            except OSError:
                pass


class LocalCommunicator(object):
    """world of one: the sharded calls degenerate to the single-GPU ones"""
    rank, world = 0, 1

    def broadcast_bytes(self, payload, root=0):
        return payload

    def all_gather_rows(self, row):
        return np.asarray(row, dtype=np.float64)[None, :]


class FileCommunicator(object):
    """Both calls over a FileRendezvous: CPU rehearsals of the N > 1 path (tests), or hosts without RCCL."""

    def __init__(self, rendezvous):
        self.rdv, self.rank, self.world = rendezvous, rendezvous.rank, rendezvous.world

    def broadcast_bytes(self, payload, root=0):
        return self.rdv.all_gather(payload if self.rank == root else b"")[root]

    def all_gather_rows(self, row):
        row = np.ascontiguousarray(row, dtype=np.float64)
        return np.stack([np.frombuffer(b, dtype=np.float64) for b in self.rdv.all_gather(row.tobytes())])


def open_communicator(ctx, rank, world, rendezvous=None, transport="rccl"):
    """The communicator of a rank of the sharded seam -- the driver (rank 0) and every worker call this with the same arguments.
    transport "rccl": MgCommunicator; its set-up is a collective that has to succeed on EVERY rank or on none, so the ranks tell
    each other how it went through the rendezvous, and if any rank failed (no librccl, a device RCCL refuses, ...) all of them
    finalise and carry the exchange through files instead (FileCommunicator: same results, the payloads are a few hundred bytes
    per step), with a warning that says why.  transport "files": FileCommunicator."""
    rank, world = int(rank), int(world)
    if world == 1:
        return LocalCommunicator()
    if transport == "files":
        return FileCommunicator(rendezvous)
    uid, err = b"", b""
    if rank == 0:
        try:
            uid = ctx.dist_unique_id()
        except Exception as e:   # noqa: BLE001 -- whatever it was, the other ranks must hear of it
            err = ("rank 0: mg_dist_unique_id: %s" % e).encode()[:300]
    uid, err0 = rendezvous.all_gather(uid)[0], rendezvous.all_gather(err)[0]
    comm = None
    if not err0:
        try:
            comm = MgCommunicator(ctx, rank, world, unique_id=uid)
        except Exception as e:   # noqa: BLE001
            err = ("rank %d: mg_dist_init: %s" % (rank, e)).encode()[:300]
        errs = [e for e in rendezvous.all_gather(err) if e]
        err0 = errs[0] if errs else b""
    if err0:
        try:
            ctx.dist_finalize()
        except Exception:   # noqa: BLE001
            pass
        import warnings
        warnings.warn("RCCL communicator could not be set up (%s): the ranks exchange through files instead" % err0.decode(errors="replace"))
        return FileCommunicator(rendezvous)
    return comm


class MgCommunicator(object):
    """Both calls through libmg_hip's RCCL entry points on the context's stream (mg_dist_broadcast, mg_dist_all_gather): what
    runs on a node of MI355X over xGMI.  The unique id travels through `rendezvous` (any object with all_gather(bytes))."""

    def __init__(self, ctx, rank, world, rendezvous=None, unique_id=None):
        self.ctx, self.rank, self.world = ctx, int(rank), int(world)
        uid = unique_id
        if uid is None:
            uid = ctx.dist_unique_id() if self.rank == 0 else b""
            if self.world > 1:
                uid = rendezvous.all_gather(uid)[0]
        ctx.dist_init(self.rank, self.world, uid)
        self._stage = None

    def _staging(self, nbytes):
        if self._stage is None or self._stage.nbytes < nbytes:
            if self._stage is not None:
                self._stage.free()
            self._stage = self.ctx.malloc(max(int(nbytes), 4096))
        return self._stage

    def broadcast_bytes(self, payload, root=0):
        head = np.array([len(payload) if self.rank == root else 0], dtype=np.int64)
        d = self._staging(8)
        if self.rank == root:
            self.ctx.upload_into(d, head)
        self.ctx.dist_broadcast(d, 8, root)
        n = int(self.ctx.download(d, (1,), np.int64)[0])
        d = self._staging(n)
        if self.rank == root:
            self.ctx.upload_into(d, np.frombuffer(payload, dtype=np.uint8))
        self.ctx.dist_broadcast(d, n, root)
        return payload if self.rank == root else self.ctx.download(d, (n,), np.uint8).tobytes()

    def all_gather_rows(self, row):
        row = np.ascontiguousarray(row, dtype=np.float64)
        d_local = self.ctx.upload(row)
        d_all = self.ctx.malloc(self.world * row.nbytes)
        try:
            self.ctx.dist_all_gather(d_local, d_all, row.size, np.float64)
            return self.ctx.download(d_all, (self.world, row.size), np.float64)
        finally:
            d_local.free()
            d_all.free()

    def close(self):
        if self._stage is not None:
            self._stage.free()
            self._stage = None
        self.ctx.dist_finalize()


def combine_first_minimum(rows):
    """rows[r] = [error, global index, latent ...] of rank r's first minimum (error +inf: nothing to offer).  The winner by
    (smaller error, then smaller global index): the first-minimum rule over the global candidate order, whatever the
    rank order.  Returns (index, error, latent) -- (0, inf, latent of rank 0) when nothing wins."""
    rows = np.asarray(rows, dtype=np.float64)
    best = None
    for r in range(rows.shape[0]):
        v, i = rows[r, 0], rows[r, 1]
        if not (v < np.inf):          # NaN and +inf never win
            continue
        if best is None or v < rows[best, 0] or (v == rows[best, 0] and i < rows[best, 1]):
            best = r
    if best is None:
        return 0, float("inf"), rows[0, 2:].copy()
    return int(rows[best, 1]), float(rows[best, 0]), rows[best, 2:].copy()


def sharded_first_minimum(comm, n_total, width, score_block):
    """The data path of every sharded step.  score_block(begin, end) -> (local_index, error, latent[width]) of the first minimum
    among the global rows [begin, end) (local_index relative to begin; error +inf if nothing wins).  One all-gather of
    2 + width float64 per rank; every rank returns the same (global index, error, latent)."""
    b, e = shard_range(n_total, comm.rank, comm.world)
    row = np.zeros(2 + int(width))
    row[0], row[1] = np.inf, float(b)
    if e > b:
        li, err, latent = score_block(b, e)
        row[0] = err if err < np.inf else np.inf
        row[1] = float(b + li)
        row[2:] = np.asarray(latent, dtype=np.float64).ravel()[:int(width)]
    return combine_first_minimum(comm.all_gather_rows(row))


# ---- commands: what rank 0 broadcasts, what every rank (rank 0 included) then executes ------------------------------
# `nodes`: {key: primitive or graph node} -- every rank holds the same primitives under the same keys -- plus
# "__skeleton__" (the rank's own _capi.Skeleton, when constraints need one) and "__primitive_set__" (planner steps).
# Constraints travel in device form (lists of plain dicts) with the alignment record rank 0 derived from the previous frames.
def _cmd_evaluate_samples(comm, nodes, cmd, scorer=None):
    """evaluate_samples_using_constraints over the ranks: the candidates came from the host (sklearn's stream) and travel
    with the command; every rank scores its block, first minimum over all.  scorer(node, device_form, alignment, block,
    skeleton) -> (index, error): the CPU rehearsal's stand-in for the HIP scorer."""
    samples = np.asarray(cmd["samples"])
    node = nodes[cmd["node"]]
    skeleton = nodes.get("__skeleton__") if cmd.get("skeleton") else None
    if scorer is None:
        from .candidate_scoring import first_minimum_of_block as scorer

    def block(b, e):
        li, err = scorer(node, cmd["constraints"], cmd["alignment"], samples[b:e], skeleton)
        return li, err, samples[b + li]
    return sharded_first_minimum(comm, len(samples), samples.shape[1], block)


def _cmd_sample_and_evaluate(comm, nodes, cmd, sampler=None):
    """gpu_batch with device sampling over the ranks: the command carries the seed and the component counts rank 0 drew from
    NumPy's stream; every rank draws ITS rows of that one draw (mg_gmm_sample_rows: the union is the single-GPU draw),
    scores them and offers its first minimum.  sampler(node, device_form, alignment, counts, seed, begin, end, skeleton,
    dtype) -> (index, error, latent): the CPU rehearsal's stand-in."""
    node = nodes[cmd["node"]]
    skeleton = nodes.get("__skeleton__") if cmd.get("skeleton") else None
    n = int(np.sum(cmd["counts"]))
    if sampler is None:
        from .candidate_scoring import sample_rows_and_first_minimum as sampler
    width = cmd.get("width")
    if width is None:
        prim_obj = node.motion_primitive if hasattr(node, "motion_primitive") else node
        width = prim_obj._prim.n_gmm_dims

    def block(b, e):
        return sampler(node, cmd["constraints"], cmd["alignment"], cmd["counts"], cmd["seed"], b, e, skeleton, np.dtype(cmd.get("dtype", "float32")))
    return sharded_first_minimum(comm, n, width, block)


def _cmd_options_step(comm, nodes, cmd, stepper=None):
    """A planner step over the ranks: every rank runs mg_options_step_rows on its block of every option's draw; one all-gather
    of all options' {error, global index, latent} records; per option the first minimum over the ranks.
    stepper(cmd, begin, end) -> {option: (global index, error, latent)}: the CPU rehearsal's stand-in."""
    options = cmd["options"]
    n = int(cmd["n_samples"])
    b, e = shard_range(n, comm.rank, comm.world)
    if stepper is None:
        pset, sk = nodes["__primitive_set__"], (nodes.get("__skeleton__") if cmd.get("skeleton") else None)
        stepper = lambda c, bb, ee: pset.options_step_rows(c, bb, ee, skeleton=sk)   # noqa: E731
    recs = stepper(cmd, b, e) if e > b else {}
    width = 2 + max(int(cmd["widths"][o]) for o in options)
    flat = np.zeros(len(options) * width)
    for k, o in enumerate(options):
        flat[k * width], flat[k * width + 1] = np.inf, float(b)
        if o in recs:
            gi, err, lat = recs[o]
            flat[k * width] = err if err < np.inf else np.inf
            flat[k * width + 1] = gi
            flat[k * width + 2:k * width + 2 + len(lat)] = lat
    everyone = comm.all_gather_rows(flat).reshape(comm.world, len(options), width)
    out = {}
    for k, o in enumerate(options):
        gi, err, lat = combine_first_minimum(everyone[:, k, :])
        out[o] = (lat[:int(cmd["widths"][o])], err, gi)
    return out


COMMANDS = {"evaluate_samples": _cmd_evaluate_samples, "sample_and_evaluate": _cmd_sample_and_evaluate, "options_step": _cmd_options_step}


def run_command(comm, nodes, cmd=None, **hooks):
    """Rank 0 passes the command; the other ranks pass None and receive it.  Every rank returns the command's result
    (the same on all ranks), or None for {"op": "stop"}."""
    payload = comm.broadcast_bytes(pickle.dumps(cmd, protocol=4) if comm.rank == 0 else b"", 0)
    cmd = pickle.loads(payload)
    if cmd.get("op") == "stop":
        return None
    return COMMANDS[cmd["op"]](comm, nodes, cmd, **hooks)


def worker_loop(comm, nodes, **hooks):
    """What ranks > 0 run: execute the commands rank 0 broadcasts until {"op": "stop"}; returns how many were served."""
    served = 0
    while run_command(comm, nodes, None, **hooks) is not None:
        served += 1
    return served


def stop_workers(comm):
    """rank 0: end the workers' loops"""
    run_command(comm, {}, {"op": "stop"})
