"""The product seam on several ranks: rank 0 drives, the other ranks sit in distributed.worker_loop.

CPU tests: world size 2 (and 3, ragged) over the file transport with the oracle as scorer -- evaluate_samples, the
device-sampling step (a seeded NumPy stand-in for the counter-based sampler: rows of ONE global draw) and the planner step give
the single-process winner.  GPU tests: a rank's rows of the device draw are bit for bit the rows of the whole draw; a planner
step in two blocks equals the unsharded step; two processes on the one GPU of the test box, files as transport and the HIP
scorers underneath, pick the single-process winner; MgCommunicator (RCCL) with the one rank a test box has."""
import multiprocessing as mp
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


# ---- CPU stand-ins for the three HIP legs (module level: the spawned ranks import them) -----------------------------------
def _model():
    from morphablegraphs_amd import synthetic
    return synthetic.make_primitive(seed=4, n_components=8, n_frames=30, n_dim=11, n_gmm=2, name="p")


def _oracle_errors(data, S, device_form):
    from oracle import c_oracle
    cp = c_oracle.COraclePrimitive(data)
    rows = []
    for c in device_form:
        tgt = [np.nan if v is None else float(v) for v in c["target"]]
        if c["type"] == "position":
            rows.append([0, c["t"], c["weight"], tgt[0], tgt[1], tgt[2], 0, 0])
        else:
            rows.append([1, c["t"], c["weight"], tgt[0], tgt[1], 0.0, 0.0, 1.0])
    return cp.keyframe_errors_f64(np.asarray(S, dtype=np.float64), np.array(rows, dtype=np.float64))


def _global_draw(counts, seed, width):
    n = int(np.sum(counts))
    return np.random.default_rng(int(seed)).standard_normal((n, width))   # stand-in: one global draw, any rows of it


def cpu_scorer(node, device_form, alignment, block, skeleton):
    from morphablegraphs_amd.distributed import first_min_argmin
    return first_min_argmin(_oracle_errors(node, block, device_form))


def cpu_sampler(node, device_form, alignment, counts, seed, b, e, skeleton, dtype):
    from morphablegraphs_amd.distributed import first_min_argmin
    X = _global_draw(counts, seed, 8)[b:e]
    li, err = first_min_argmin(_oracle_errors(node, X, device_form))
    return li, err, X[li]


def cpu_stepper(cmd, b, e):
    from morphablegraphs_amd.distributed import first_min_argmin
    out = {}
    for k, o in enumerate(cmd["options"]):
        X = _global_draw(cmd["counts"][o], cmd["seed"] + k, 8)[b:e]
        li, err = first_min_argmin(_oracle_errors(_model(), X, cmd["constraints"][o]))
        out[o] = (b + li, err, X[li])
    return out


CONS = [{"type": "position", "t": 29.0, "weight": 1.0, "target": [10.0, None, -20.0]},
        {"type": "direction", "t": 29.0, "weight": 2.0, "target": [0.3, -1.0]}]


def _commands(n):
    rng = np.random.default_rng(7)
    S = rng.standard_normal((n, 8))
    j = int(np.argmin(_oracle_errors(_model(), S, CONS)))
    S[(j + n // 2) % n] = S[j]            # the minimum twice, in different ranks' blocks: the FIRST of the two must win
    counts = np.array([n - n // 3, n // 3], dtype=np.int64)
    return [{"op": "evaluate_samples", "node": "p", "samples": S, "constraints": CONS, "alignment": None, "skeleton": False},
            {"op": "sample_and_evaluate", "node": "p", "constraints": CONS, "alignment": None, "skeleton": False, "counts": counts, "seed": 11,
             "dtype": "float64", "width": 8},
            {"op": "options_step", "options": ["a", "b", "c"], "n_samples": n, "seed": 5, "dtype": "float64", "skeleton": False,
             "counts": {o: counts for o in "abc"}, "constraints": {o: CONS for o in "abc"}, "alignments": {o: None for o in "abc"},
             "widths": {o: 8 for o in "abc"}}]


def _cpu_rank(rank, world, base, n, out_dir):
    sys.path.insert(0, ROOT)
    from morphablegraphs_amd import distributed
    comm = distributed.FileCommunicator(distributed.FileRendezvous(rank, world, base=base, timeout=60.0))
    nodes = {"p": _model()}
    if rank == 0:
        res = []
        for cmd in _commands(n):
            h = {"evaluate_samples": {"scorer": cpu_scorer}, "sample_and_evaluate": {"sampler": cpu_sampler}, "options_step": {"stepper": cpu_stepper}}[cmd["op"]]
            res.append(distributed.run_command(comm, nodes, cmd, **h))
        distributed.stop_workers(comm)
        import pickle
        with open(os.path.join(out_dir, "r0.pkl"), "wb") as f:
            pickle.dump(res, f)
    else:
        # a worker does not know which command comes: it passes the hook each command understands
        served = 0
        while True:
            import pickle
            payload = comm.broadcast_bytes(b"", 0)
            cmd = distributed.decode_command(payload)
            if cmd["op"] == "stop":
                break
            h = {"evaluate_samples": {"scorer": cpu_scorer}, "sample_and_evaluate": {"sampler": cpu_sampler}, "options_step": {"stepper": cpu_stepper}}[cmd["op"]]
            distributed.COMMANDS[cmd["op"]](comm, nodes, cmd, **h)
            served += 1
        with open(os.path.join(out_dir, "served%d" % rank), "w") as f:
            f.write(str(served))


@pytest.mark.parametrize("world,n", [(2, 64), (2, 37), (3, 50)])
def test_sharded_commands_over_files_equal_the_single_process(tmp_path, world, n):
    import pickle
    from morphablegraphs_amd import distributed
    base = str(tmp_path / "rdv")
    ctx = mp.get_context("spawn")
    procs = [ctx.Process(target=_cpu_rank, args=(r, world, base, n, str(tmp_path))) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    with open(tmp_path / "r0.pkl", "rb") as f:
        sharded = pickle.load(f)
    for r in range(1, world):
        assert (tmp_path / ("served%d" % r)).read_text() == "3"
    # the same commands in one process
    local = distributed.LocalCommunicator()
    nodes = {"p": _model()}
    cmds = _commands(n)
    one = [distributed._cmd_evaluate_samples(local, nodes, cmds[0], scorer=cpu_scorer),
           distributed._cmd_sample_and_evaluate(local, nodes, cmds[1], sampler=cpu_sampler),
           distributed._cmd_options_step(local, nodes, cmds[2], stepper=cpu_stepper)]
    for k in (0, 1):
        assert sharded[k][0] == one[k][0] and sharded[k][1] == one[k][1]
        np.testing.assert_array_equal(sharded[k][2], one[k][2])
    for o in "abc":
        assert sharded[2][o][1] == one[2][o][1] and sharded[2][o][2] == one[2][o][2]
        np.testing.assert_array_equal(sharded[2][o][0], one[2][o][0])
    # and the reference's loop over ALL candidates finds that winner (first strict minimum)
    ref = _oracle_errors(_model(), cmds[0]["samples"], CONS)
    best = 0
    for i, e in enumerate(ref):
        if ref[best] > e:
            best = i
    assert sharded[0][0] == best and sharded[0][1] == ref[best]
    dups = np.flatnonzero(ref == ref[best])
    assert len(dups) == 2 and best == dups[0]            # the tie was there and its first row won


# ---- GPU ---------------------------------------------------------------------------------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("n_gmm,force_valu", [(8, 0), (8, 1), (17, 0)])
def test_a_ranks_rows_are_the_rows_of_the_whole_draw(n_gmm, force_valu):
    """mg_gmm_sample_rows: any block of rows of the draw, bit for bit (matrix-pipe sampler, lane-per-row sampler, prefix sums
    staged on the device for more than 16 components), for blocks that cut tiles and components."""
    from morphablegraphs_amd import _capi, synthetic
    ctx = _capi.Context(0)
    data = synthetic.make_primitive(seed=21, n_components=24, n_frames=60, n_gmm=n_gmm, name="p")
    prim = _capi.Primitive(ctx, data)
    ctx.set_option(_capi.MG_OPT_FORCE_VALU_SAMPLE, force_valu)
    rng = np.random.default_rng(3)
    n = 1000
    counts = rng.multinomial(n, np.full(n_gmm, 1.0 / n_gmm)).astype(np.int64)
    for dtype in (np.float32, np.float64):
        X, comp = prim.gmm_sample(counts, 99, dtype=dtype)
        L = prim.n_gmm_dims
        for b, e in ((0, n), (0, 1), (n - 1, n), (0, 333), (333, 1000), (17, 18), (250, 750), (499, 516)):
            d_x, d_c = ctx.malloc((e - b) * L * np.dtype(dtype).itemsize), ctx.malloc((e - b) * 4)
            prim.gmm_sample_dev(counts, 99, d_x, dtype, L, component_dev=d_c, rows=(b, e - b))
            np.testing.assert_array_equal(ctx.download(d_x, (e - b, L), dtype).view(np.uint8), X[b:e].view(np.uint8))
            np.testing.assert_array_equal(ctx.download(d_c, (e - b,), np.int32), comp[b:e])
            d_x.free()
            d_c.free()
    prim.close()
    ctx.close()


@pytest.mark.gpu
def test_a_planner_step_in_blocks_is_the_unsharded_step():
    """mg_options_step_rows on [0, n/3), [n/3, n) (one launch each, and the per-option chains): the candidates and errors of
    the blocks are the unsharded step's, and the blocks' records combine to its winners."""
    from morphablegraphs_amd import _capi, distributed, synthetic
    from morphablegraphs_amd.candidate_scoring import constraints_to_device_form
    from morphablegraphs_amd.motion_state_graph import HipPrimitiveSet
    prims = synthetic.make_graph_primitives(5)
    names = [p["name"] for p in prims]
    pset = HipPrimitiveSet(prims)
    n = 1500
    cons = {nm: [{"type": "position", "t": float(p["n_canonical_frames"] - 1), "weight": 1.0, "target": [10.0, None, 5.0]},
                 {"type": "direction", "t": float(p["n_canonical_frames"] - 1), "weight": 1.0, "target": [0.5, 1.0]}] for nm, p in zip(names, prims)}
    rng = np.random.default_rng(1)
    cmd = {"op": "options_step", "options": names, "n_samples": n, "seed": 77, "dtype": "float32", "skeleton": False,
           "counts": {}, "constraints": {nm: constraints_to_device_form(cons[nm]) for nm in names}, "alignments": {nm: None for nm in names},
           "widths": {nm: pset.nodes[nm]._prim.n_gmm_dims for nm in names}}
    for nm, p in zip(names, prims):
        w = np.asarray(p["gmm_weights"], dtype=np.float64)
        cmd["counts"][nm] = rng.multinomial(n, w / w.sum()).astype(np.int64)
    for mode in (0, 1):
        pset.ctx.set_option(_capi.MG_OPT_OPTIONS_STEP, mode)
        whole = pset.options_step_rows(cmd, 0, n)
        plan = pset._step_plan(tuple(names), n, np.dtype(np.float32))
        X = {st[0]: st[3].download(st[4], (n, st[7]), np.float32) for st in plan["steps"]}
        E = {st[0]: st[3].download(st[5], (n,), np.float64) for st in plan["steps"]}
        parts = []
        for b, e in ((0, n // 3), (n // 3, n)):
            parts.append(pset.options_step_rows(cmd, b, e))
            plan_b = pset._step_plan(tuple(names), e - b, np.dtype(np.float32))
            for st in plan_b["steps"]:
                np.testing.assert_array_equal(st[3].download(st[4], (e - b, st[7]), np.float32).view(np.uint32), X[st[0]][b:e].view(np.uint32))
                np.testing.assert_array_equal(st[3].download(st[5], (e - b,), np.float64).view(np.uint64), E[st[0]][b:e].view(np.uint64))
        for nm in names:
            rows = np.stack([np.concatenate([[parts[r][nm][1], parts[r][nm][0]], parts[r][nm][2]]) for r in range(2)])
            gi, err, lat = distributed.combine_first_minimum(rows)
            assert gi == whole[nm][0] and err == whole[nm][1]
            np.testing.assert_array_equal(lat[:len(whole[nm][2])], whole[nm][2])
            assert gi == int(np.argmin(E[nm])) and err == E[nm][gi]
    pset.ctx.set_option(_capi.MG_OPT_OPTIONS_STEP, 0)


def _gpu_rank(rank, world, base, out_dir, transport="files"):
    """a rank of the sharded seam on the test box's one GPU: HIP scorers, files as transport (asked for, or fallen back to)"""
    sys.path.insert(0, ROOT)
    import pickle
    from morphablegraphs_amd import _capi, distributed, synthetic
    from morphablegraphs_amd.candidate_scoring import evaluate_samples_using_constraints, sample_and_evaluate_on_device
    from morphablegraphs_amd.motion_state_graph import HipMotionStateGraphNode, HipPrimitiveSet
    ctx = _capi.Context(0)
    import warnings
    with warnings.catch_warnings(record=True) as caught:
        warnings.simplefilter("always")
        comm = distributed.open_communicator(ctx, rank, world, distributed.FileRendezvous(rank, world, base=base, timeout=120.0), transport=transport)
    if transport == "rccl":   # two ranks on one device: RCCL refuses, every rank must have fallen back
        assert isinstance(comm, distributed.FileCommunicator) and any("RCCL communicator could not be set up" in str(w.message) for w in caught)
    node = HipMotionStateGraphNode(context=ctx)
    node.init_from_dict("walk", {"name": "leftStance", "mm": synthetic.make_walk_primitive(seed=0)})
    prims = synthetic.make_graph_primitives(4)
    pset = HipPrimitiveSet(prims, context=ctx)
    nodes = {node.node_key: node, "__primitive_set__": pset}
    nodes.update(pset.nodes)
    if rank != 0:
        served = distributed.worker_loop(comm, nodes)
        with open(os.path.join(out_dir, "served%d" % rank), "w") as f:
            f.write(str(served))
        return
    cons = [{"type": "position", "t": 155.0, "weight": 1.0, "target": [40.0, None, -30.0]},
            {"type": "direction", "t": 155.0, "weight": 1.0, "target": [0.5, 1.0]}]
    names = [p["name"] for p in prims]
    ocons = {nm: [{"type": "position", "t": float(p["n_canonical_frames"] - 1), "weight": 1.0, "target": [10.0, None, 5.0]}] for nm, p in zip(names, prims)}
    # ... and a step in which one option carries a per-frame constraint (the root never closer than ... to a point): option by option
    fcons = dict(ocons)
    fcons[names[2]] = ocons[names[2]] + [{"type": "frame_ca_position", "joint": "root", "target": [3.0, None, -2.0],
                                          "n_frames": prims[2]["n_canonical_frames"], "weight": 2.0}]
    out = {}
    for label, c in (("sharded", comm), ("single", None)):
        np.random.seed(5)
        S = node.motion_primitive.sample_low_dimensional_vector(1001)
        out[label] = [evaluate_samples_using_constraints(S, node, cons, communicator=c),
                      sample_and_evaluate_on_device(node, cons, 3001, seed=9, communicator=c),
                      pset.evaluate_options_on_device(names, ocons, n_samples=2049, seed=3, communicator=c),
                      pset.evaluate_options_on_device(names, fcons, n_samples=1025, seed=11, communicator=c)]
    distributed.stop_workers(comm)
    with open(os.path.join(out_dir, "r0.pkl"), "wb") as f:
        pickle.dump(out, f)


@pytest.mark.gpu
@pytest.mark.parametrize("transport", ["files", "rccl"])
def test_two_ranks_on_one_gpu_pick_the_single_process_winner(tmp_path, transport):
    """Rank 0 drives through the product entry points with a communicator, rank 1 sits in worker_loop; both use the HIP library
    on the box's one GPU, the exchange goes through files -- asked for, or ("rccl") after the RCCL set-up failed on the shared
    device and both ranks fell back.  Same winners, same errors, bit for bit, as without a communicator."""
    import pickle
    base = str(tmp_path / "rdv")
    ctx = mp.get_context("spawn")
    procs = [ctx.Process(target=_gpu_rank, args=(r, 2, base, str(tmp_path), transport)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(300)
        assert p.exitcode == 0
    assert (tmp_path / "served1").read_text() == "7"          # three commands + one per option of the step with a per-frame constraint
    with open(tmp_path / "r0.pkl", "rb") as f:
        out = pickle.load(f)
    for k in (0, 1):
        np.testing.assert_array_equal(out["sharded"][k][0], out["single"][k][0])
        assert out["sharded"][k][1] == out["single"][k][1]
    for k in (2, 3):
        assert out["sharded"][k][0] == out["single"][k][0]
        for nm, (lat, err) in out["single"][k][1].items():
            np.testing.assert_array_equal(out["sharded"][k][1][nm][0], lat)
            assert out["sharded"][k][1][nm][1] == err


@pytest.mark.gpu
def test_rccl_communicator_with_the_one_rank_a_test_box_has():
    """MgCommunicator: mg_dist_init / mg_dist_broadcast / mg_dist_all_gather through RCCL with world size 1 (more ranks need more
    GPUs than a test box has), and a command through it."""
    from morphablegraphs_amd import _capi, distributed, synthetic
    from morphablegraphs_amd.motion_state_graph import HipMotionStateGraphNode
    ctx = _capi.Context(0)
    comm = distributed.MgCommunicator(ctx, 0, 1)
    payload = bytes(range(256)) * 37
    assert comm.broadcast_bytes(payload, 0) == payload
    row = np.arange(43, dtype=np.float64) * 0.5
    np.testing.assert_array_equal(comm.all_gather_rows(row), row[None, :])
    node = HipMotionStateGraphNode(context=ctx)
    node.init_from_dict("walk", {"name": "leftStance", "mm": synthetic.make_walk_primitive(seed=0)})
    cons = [{"type": "position", "t": 155.0, "weight": 1.0, "target": [40.0, None, -30.0]}]
    counts = np.array([100, 50, 150, 0, 25, 75, 60, 40], dtype=np.int64)
    cmd = {"op": "sample_and_evaluate", "node": node.node_key, "constraints": cons, "alignment": None, "skeleton": False, "counts": counts,
           "seed": 4, "dtype": "float32"}
    gi, err, lat = distributed.run_command(comm, {node.node_key: node}, cmd)
    from morphablegraphs_amd.candidate_scoring import sample_rows_and_first_minimum
    li, err1, lat1 = sample_rows_and_first_minimum(node, cons, None, counts, 4, 0, 500)
    assert gi == li and err == err1
    np.testing.assert_array_equal(lat, lat1)
    comm.close()
    ctx.close()
