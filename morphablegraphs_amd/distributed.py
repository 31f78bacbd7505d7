"""Candidate-batch sharding across the GPUs of one node.

Candidates are independent units: rank r owns the contiguous block [r*B/N, (r+1)*B/N), the per-primitive
constants are replicated, and the only exchange is one all-gather of the per-rank scores (RCCL over xGMI
on GPUs, gloo in the CPU tests) followed by the reference's first-minimum argmin over the GLOBAL index order
(reference motion_primitive_generator.py:251-257).  The carrier is the library's own RCCL entry points
(mg_all_gather_scores / mg_sharded_best_candidate over a _capi.Context after dist_init; MgCommunicator for the product seam) or,
for CPU rehearsals and ranks that share a GPU, files (FileCommunicator).  This package imports no tensor framework: the
framework-collective rehearsal of the same exchange lives with the tests (tests/test_distributed_gloo.py injects a gloo-backed
communicator into THIS module's commands).
The scores come from the scorer callable (libmg_hip on a GPU box).
"""
import json
import os
import stat
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


def mg_all_gather_scores(ctx, local_scores, n_total, rank, world, dtype=np.float64):
    """The all-gather of per-rank score blocks through mg_dist_all_gather (RCCL loaded by libmg_hip.so, communicator
    made by ctx.dist_init): no tensor framework anywhere.  Blocks of different lengths are padded with +inf to the longest."""
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
    """The sharded argmin on the library's own collective: every rank scores its contiguous block with `scorer(block)` and the
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
# (FileCommunicator: CPU rehearsals and tests, no GPU) and in one process (LocalCommunicator).
# ---------------------------------------------------------------------------------------------------------------------
def private_rendezvous_dir(name):
    """A directory only this user can enter, under the temporary directory: created 0700, or -- if it exists -- checked to belong
    to this user and to be closed to everybody else (another local user cannot plant files the ranks would read).  The ranks
    of a run share it by NAME (launcher's pid + master port); a launcher creates it with tempfile.mkdtemp and hands the path
    down (MG_RDV_DIR)."""
    path = os.path.join(tempfile.gettempdir(), name)
    try:
        os.mkdir(path, 0o700)
    except FileExistsError:
        pass
    st = os.lstat(path)
    if not stat.S_ISDIR(st.st_mode) or st.st_uid != os.getuid() or (st.st_mode & 0o077):
        raise PermissionError("rendezvous directory %s is not a private directory of this user" % path)
    return path


class FileRendezvous(object):
    """Bytes between the ranks of one node: files in a PRIVATE directory (0700, owned by this user), written with an
    atomic rename.  Carries the 128-byte RCCL unique id to the ranks (bench.py, worker processes) and, in CPU rehearsals,
    everything else.  base: a path prefix inside such a directory (default: $MG_RDV_DIR/rdv, else a directory named after the
    launcher's pid and the master port); a run's leftovers under the same prefix are removed when rank 0 starts."""

    def __init__(self, rank, world, base=None, timeout=300.0):
        self.rank, self.world, self.timeout = int(rank), int(world), float(timeout)
        if base is None:
            d = os.environ.get("MG_RDV_DIR") or private_rendezvous_dir("mg_rdv_%d_%d_%s" % (os.getuid(), os.getppid(), os.environ.get("MASTER_PORT", "0")))
            base = os.path.join(d, "rdv")
        self.base = base
        self.seq = 0
        self.cleanup()      # what a run that ended early left behind under this rank's names (give every run a base of its own)

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
    # Preflight: what can fail on ONE rank before the collective set-up (no librccl, an unusable device) is found out and told to
    # the others first -- a rank that entered ncclCommInitRank alone would sit there for good.
    uid, err = b"", b""
    try:
        ctx.dist_preflight()
        if rank == 0:
            uid = ctx.dist_unique_id()
    except Exception as e:   # noqa: BLE001 -- whatever it was, the other ranks must hear of it
        err = ("rank %d: %s" % (rank, e)).encode()[:300]
    errs = [e for e in rendezvous.all_gather(err) if e]
    uid, err0 = rendezvous.all_gather(uid)[0], (errs[0] if errs else b"")
    err = b""
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


def encode_command(cmd):
    """A command as bytes without pickle: an 8-byte length, a JSON document in which every NumPy array (and NumPy scalar) has
    been replaced by a reference {"__nd__": k} / a plain number, then the arrays' raw bytes back to back.  Nothing a reader
    executes: a rank that reads a planted file gets a ValueError or wrong numbers, never code."""
    arrays = []

    def enc(v):
        if isinstance(v, np.ndarray):
            a = np.ascontiguousarray(v)
            arrays.append(a)
            return {"__nd__": len(arrays) - 1, "dtype": a.dtype.str, "shape": list(a.shape)}
        if isinstance(v, np.generic):
            return v.item()
        if isinstance(v, dict):
            return {"__dict__": [[enc(k), enc(x)] for k, x in v.items()]} if any(not isinstance(k, str) for k in v) else {k: enc(x) for k, x in v.items()}
        if isinstance(v, tuple):          # (node keys are tuples: they must come back hashable)
            return {"__tuple__": [enc(x) for x in v]}
        if isinstance(v, list):
            return [enc(x) for x in v]
        if v is None or isinstance(v, (bool, int, float, str)):
            return v
        raise TypeError("a command may hold numbers, strings, lists, dicts and arrays, not %r" % type(v).__name__)
    doc = json.dumps(enc(cmd), allow_nan=True).encode("utf-8")
    return len(doc).to_bytes(8, "little") + doc + b"".join(a.tobytes() for a in arrays)


def decode_command(payload):
    n = int.from_bytes(payload[:8], "little")
    doc = json.loads(payload[8:8 + n].decode("utf-8"))
    blob, pos = memoryview(payload)[8 + n:], [0]
    arrays = {}

    def dec(v):
        if isinstance(v, dict):
            if "__nd__" in v:
                dt = np.dtype(v["dtype"])
                if dt.hasobject:
                    raise ValueError("object arrays do not travel")
                count = int(np.prod(v["shape"])) if v["shape"] else 1
                k = v["__nd__"]
                if k not in arrays:                     # arrays are referenced in the order they were appended
                    if k != len(arrays):
                        raise ValueError("malformed command")
                    arrays[k] = np.frombuffer(blob, dtype=dt, count=count, offset=pos[0]).reshape(v["shape"]).copy()
                    pos[0] += count * dt.itemsize
                return arrays[k]
            if "__tuple__" in v:
                return tuple(dec(x) for x in v["__tuple__"])
            if "__dict__" in v:
                return {dec(k): dec(x) for k, x in v["__dict__"]}
            return {k: dec(x) for k, x in v.items()}
        if isinstance(v, list):
            return [dec(x) for x in v]
        return v
    return dec(doc)


def run_command(comm, nodes, cmd=None, **hooks):
    """Rank 0 passes the command; the other ranks pass None and receive it.  Every rank returns the command's result
    (the same on all ranks), or None for {"op": "stop"}."""
    payload = comm.broadcast_bytes(encode_command(cmd) if comm.rank == 0 else b"", 0)
    cmd = decode_command(payload)
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
