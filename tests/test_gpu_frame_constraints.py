"""GPU tests of the constraints that walk a joint through EVERY frame of a candidate (morphablegraphs_amd/frame_constraints.py):
trajectory constraints on other joints than the root, collision-avoidance positions, discrete / local trajectories, trajectory
sets and the local joint rotation -- device tracks + host arithmetic against the oracle's frame-by-frame restatements of the
reference classes (oracle/mg_oracle.py per_frame_constraint_residuals), in local coordinates, aligned to a previous frame and to
a start pose, and through the reference-shaped entry points (evaluate_samples_using_constraints, the sample filter, the
optimiser's objectives)."""
import numpy as np
import pytest

from morphablegraphs_amd import _capi, synthetic
from morphablegraphs_amd.candidate_scoring import (HipSampleFilter, alignment_from_start_pose, errors_of_samples,
                                                   evaluate_samples_using_constraints)
from morphablegraphs_amd.frame_constraints import frame_constraint_residuals, frame_constraints_errors
from test_gpu_adaptors import _path_following_model, _primitive

pytestmark = pytest.mark.gpu


def _setup(n=9, seed=21):
    from oracle import mg_oracle as orc
    data = _path_following_model()
    mp = _primitive(data)
    op = orc.OraclePrimitive(data)
    joints, animated = synthetic.make_skeleton()
    sk = _capi.Skeleton(joints, animated)
    S = np.random.default_rng(seed).standard_normal((n, 40))
    return orc, data, mp, op, joints, animated, sk, S


def _constraints(op, orc, S, joints, animated):
    """One constraint of every per-frame type near the candidates' own motions (targets from candidate 0's joints)."""
    frames0 = op.back_project_frames(S[0])
    hand = np.array([orc.joint_global_position(f, joints, animated, "LeftHand") for f in frames0])
    foot = np.array([orc.joint_global_position(f, joints, animated, "RightFoot") for f in frames0])
    hand_cps = hand[::22].copy() + np.array([1.0, -0.5, 2.0])
    foot_cps = foot[::30].copy() + np.array([-1.0, 0.3, 1.0])
    F = op.n_canonical_frames
    return [
        {"type": "frame_joint_trajectory", "joint": "LeftHand", "control_points": hand_cps.tolist(), "min_u": 0.05, "granularity": 1000, "weight": 1.5},
        {"type": "frame_ca_position", "joint": "RightFoot", "target": [float(foot[60, 0]) + 3.0, None, float(foot[60, 2]) - 2.0], "n_frames": F, "weight": 2.0},
        {"type": "frame_discrete_trajectory", "joint": "LeftHand", "points": (hand[:120] + 0.7).tolist(), "unconstrained": [1], "weight": 0.8},
        {"type": "frame_local_trajectory", "joint": "Hips", "control_points": (frames0[::20, :3] + np.array([0.5, 0.0, -0.5])).tolist(), "granularity": 1000,
         "start_t": 3.0, "n_frames": F, "weight": 1.2},
        {"type": "frame_trajectory_set", "joints": ["LeftHand", "RightFoot"],
         "trajectories": [{"control_points": hand_cps.tolist(), "granularity": 1000, "range_start": 20.0, "range_end": 1500.0},
                          {"control_points": foot_cps.tolist(), "granularity": 1000, "range_start": None, "range_end": None}],
         "arc_lengths": [4.0, 1.0], "n_frames": F, "weight": 1.1},
        {"type": "frame_joint_rotation", "joint_index": 7, "quaternion": [0.9, 0.1, -0.3, 0.2], "frame_idx": 40.0, "weight": 0.6},
        {"type": "frame_joint_rotation", "joint_index": 0, "quaternion": [0.8, 0.0, 0.6, 0.0], "frame_idx": 12.0, "weight": 0.4},
    ]


def _check(mp, op, S, clist, sk, joints, animated, alignment, **oracle_alignment):
    total_ref, blocks_ref = op.frame_constraint_errors(S, clist, joints, animated, **oracle_alignment)
    total, blocks = frame_constraints_errors(mp._prim, S, clist, sk, alignment)
    for c, b, br in zip(clist, blocks, blocks_ref):
        assert b.shape == br.shape, (c["type"], b.shape, br.shape)
        scale = max(1.0, np.abs(br).max())
        # float64 end to end; the trajectory search carries its stated tolerance (tests/test_gpu_closest_point.py: the forward
        # difference's noise)
        tol = 2e-6 if c["type"] == "frame_joint_trajectory" else 1e-9
        assert np.abs(b - br).max() <= tol * scale, (c["type"], np.abs(b - br).max(), scale)
    loose = any(c["type"] == "frame_joint_trajectory" for c in clist)
    np.testing.assert_allclose(total, total_ref, rtol=2e-6 if loose else 1e-8, atol=1e-8)
    return total


def test_every_per_frame_constraint_in_local_coordinates():
    orc, data, mp, op, joints, animated, sk, S = _setup()
    clist = _constraints(op, orc, S, joints, animated)
    total = _check(mp, op, S, clist, sk, joints, animated, None)
    assert np.all(total > 0.0) and len(np.unique(total)) == len(total)
    # the trajectory-set term is live (its active range is met) and the inactive range alone silences it
    res, err = frame_constraint_residuals(mp._prim, S, clist[4], sk, None)
    assert np.count_nonzero(res[0]) > 10
    quiet = dict(clist[4], trajectories=[dict(t, range_start=None, range_end=None) for t in clist[4]["trajectories"]])
    assert not frame_constraint_residuals(mp._prim, S, quiet, sk, None)[0].any()


def test_per_frame_constraints_on_aligned_candidates():
    """Global coordinates: every candidate is turned and moved onto the previous motion's last frame (or a start pose) first; the
    tracks follow with the same per-candidate transform the fused scorer derives."""
    orc, data, mp, op, joints, animated, sk, S = _setup(n=6, seed=5)
    clist = _constraints(op, orc, S, joints, animated)
    prev = op.back_project_frames(np.random.default_rng(3).standard_normal(40))[-1].copy()
    prev[:3] = [25.0, 89.0, -12.0]
    for align_joint in ("Hips", "Spine1"):
        al = sk.alignment_to(prev, sk.index(align_joint))
        alignment = {"joint": sk.index(align_joint), "position": al["position"], "heading": al["heading"]}
        _check(mp, op, S, clist, sk, joints, animated, alignment, prev_frame=prev, align_joint=align_joint)
    sp = {"position": [3.0, 2.0, 1.0], "orientation": [0.0, 25.0, 0.0]}
    total_ref, blocks_ref = [], None
    # the reference's start-pose alignment rewrites the pose it is given: a fresh copy per candidate keeps candidates independent
    for b in range(len(S)):
        t, _ = op.frame_constraint_errors(S[b:b + 1], clist, joints, animated, start_pose={"position": list(sp["position"]), "orientation": list(sp["orientation"])})
        total_ref.append(t[0])
    total, _ = frame_constraints_errors(mp._prim, S, clist, sk, alignment_from_start_pose(sp))
    np.testing.assert_allclose(total, total_ref, rtol=2e-6, atol=1e-8)


def test_fused_track_scorer_is_the_chain_bit_for_bit(monkeypatch):
    """mg_joint_tracks + mg_score_frame_constraints (two launches, no frames in memory) against the chain they replace
    (mg_back_project_frames_f64 -> mg_align_frames -> mg_joint_positions -> one scorer launch per constraint): the same residual
    vectors and the same sums, BIT for bit (the verdict's bar was 1e-12), in local coordinates, aligned to a previous frame by two
    different nodes, and to a start pose; float32 and float64 latents; a list longer than one scorer launch takes (> 4)."""
    from morphablegraphs_amd import frame_constraints as fc
    orc, data, mp, op, joints, animated, sk, S = _setup(n=33, seed=4)
    clist = _constraints(op, orc, S, joints, animated)
    prev = op.back_project_frames(np.random.default_rng(3).standard_normal(40))[-1].copy()
    prev[:3] = [25.0, 89.0, -12.0]
    alignments = [None]
    for align_joint in ("Hips", "Spine1"):
        al = sk.alignment_to(prev, sk.index(align_joint))
        alignments.append({"joint": sk.index(align_joint), "position": al["position"], "heading": al["heading"]})
    alignments.append(alignment_from_start_pose({"position": [3.0, 2.0, 1.0], "orientation": [0.0, 25.0, 0.0]}))
    # the list with one more CA constraint on a third joint and other frames: 4 requests, 6 track constraints + 2 rotations
    longer = clist + [{"type": "frame_ca_position", "joint": "Head", "target": [5.0, 150.0, None], "n_frames": 57, "weight": 0.3}]
    for lat in (S, S.astype(np.float32)):
        for alignment in alignments:
            for cl in (clist, longer, clist[1:2], clist[:1]):
                monkeypatch.setattr(fc, "FUSED", True)
                total, blocks = frame_constraints_errors(mp._prim, lat, cl, sk, alignment)
                monkeypatch.setattr(fc, "FUSED", False)
                total_c, blocks_c = frame_constraints_errors(mp._prim, lat, cl, sk, alignment)
                for c, b, bc in zip(cl, blocks, blocks_c):
                    assert np.array_equal(b.view(np.uint64), bc.view(np.uint64)), (c["type"], np.abs(b - bc).max())
                assert np.array_equal(total.view(np.uint64), total_c.view(np.uint64))
    # the fused route really ran: its plan is cached, and the root alone (no skeleton) goes the same way
    assert len(fc._PLAN_CACHE) >= 4
    monkeypatch.setattr(fc, "FUSED", True)
    root = {"type": "frame_ca_position", "joint": "root", "target": [80.0, None, 20.0], "n_frames": op.n_canonical_frames, "weight": 1.0}
    a = frame_constraints_errors(mp._prim, S, [root], None, None)[0]
    monkeypatch.setattr(fc, "FUSED", False)
    assert np.array_equal(a, frame_constraints_errors(mp._prim, S, [root], None, None)[0])


def test_closest_point_walk_by_eight_lanes_is_the_one_lane_walk_bit_for_bit():
    """mg_score_trajectory / mg_score_trajectory_points give a candidate eight lanes below 65536 candidates (the grid values of the
    search side by side, a ballot for the first that does not fall) and one lane above: the same distances and the same errors,
    bit for bit -- coarse and fine grids (windows refilled: more than six grid steps per frame), paths that run past the
    spline's end, a bound in mid-spline, aligned candidates, any joint's track."""
    from morphablegraphs_amd.candidate_scoring import cached_trajectory
    orc, data, mp, op, joints, animated, sk, S = _setup(n=77, seed=8)
    prim, ctx = mp._prim, mp._prim.ctx
    frames0 = op.back_project_frames(S[0])
    prev = op.back_project_frames(np.random.default_rng(3).standard_normal(40))[-1].copy()
    prev[:3] = [25.0, 89.0, -12.0]
    al = sk.alignment_to(prev, 0)
    alignments = [None, {"joint": 0, "position": al["position"], "heading": al["heading"]},
                  alignment_from_start_pose({"position": [3.0, 2.0, 1.0], "orientation": [0.0, 25.0, 0.0]})]
    paths = [(frames0[::26, :3] + 0.25, 1000), (frames0[::26, :3] + 0.25, 7), (frames0[::26, :3] + 0.25, 20000),
             (frames0[:80:16, :3] - 1.0, 1000), (np.array([[0.0, 90.0, 0.0], [60.0, 90.0, 25.0]]), 300)]
    hand = prim.joint_tracks(sk, ["LeftHand"], S)[:, :, 0]

    def both(fn):
        out = []
        ctx.set_option(_capi.MG_OPT_TRAJECTORY_SEARCH, 1)       # (the lanes are the monotone walk's: the reference's search is one chain)
        try:
            for lanes in (0, 1, 4):       # (0: eight lanes at this batch size; 1: the streaming one-lane kernel; 4: four lanes)
                ctx.set_option(_capi.MG_OPT_TRAJECTORY_LANES, lanes)
                out.append(fn())
        finally:
            ctx.set_option(_capi.MG_OPT_TRAJECTORY_LANES, 0)
            ctx.set_option(_capi.MG_OPT_TRAJECTORY_SEARCH, 0)
        assert np.array_equal(out[2][0].view(np.uint64), out[0][0].view(np.uint64)) and np.array_equal(out[2][1].view(np.uint64), out[0][1].view(np.uint64))
        return out[:2]
    moved = 0
    for cps, gran in paths:
        traj = cached_trajectory(prim, {"type": "trajectory", "control_points": cps.tolist(), "granularity": gran})
        for alignment in alignments:
            for min_u in (0.0, 0.37):
                (e8, r8), (e1, r1) = both(lambda: prim.score_trajectory(traj, S, min_u, 1.3, alignment, residuals=True))
                assert np.array_equal(r8.view(np.uint64), r1.view(np.uint64)), (gran, min_u, np.abs(r8 - r1).max())
                assert np.array_equal(e8.view(np.uint64), e1.view(np.uint64))
                moved += int(np.count_nonzero(np.diff(r8, axis=1)))
        (e8, r8), (e1, r1) = both(lambda: prim.score_trajectory_points(traj, hand, 0.1, 0.7, residuals=True))
        assert np.array_equal(r8.view(np.uint64), r1.view(np.uint64)) and np.array_equal(e8.view(np.uint64), e1.view(np.uint64))
    assert moved > 1000


def test_trajectory_scorers_side_by_side_are_the_single_launches():
    """mg_score_trajectories: the options of a planner step, each with its own primitive, candidates and trajectory, in one launch
    -- errors equal to the single launches' bit for bit, written or added to, local, aligned to a previous frame and to a start
    pose mixed in one call, 20 scorers (more than one launch holds)."""
    from morphablegraphs_amd.motion_state_graph import HipPrimitiveSet
    prims = synthetic.make_graph_primitives(5)
    pset = HipPrimitiveSet(prims, separate_streams=False)
    ctx = pset.ctx
    for search in (0, 1):
        ctx.set_option(_capi.MG_OPT_TRAJECTORY_SEARCH, search)
        try:
            _side_by_side(pset, prims, ctx)
        finally:
            ctx.set_option(_capi.MG_OPT_TRAJECTORY_SEARCH, 0)


def _side_by_side(pset, prims, ctx):
    rng = np.random.default_rng(12)
    n = 333
    ps, trs, xs, es, lds_, min_us, ws, als, ref = [], [], [], [], [], [], [], [], []
    for i in range(20):
        p = pset.nodes[prims[i % 5]["name"]]._prim
        L = p.n_gmm_dims
        S = rng.standard_normal((n, L)).astype(np.float32)
        path = p.back_project_frames_f64(S[:1].astype(np.float64)[:, :p.n_components])[0][::9, :3] + rng.standard_normal(3)
        t = _capi.Trajectory(p, path, 500 + 100 * i)
        al = [None, {"joint": 0, "position": [1.0, 2.0, 3.0], "heading": [0.6, 0.8]}, alignment_from_start_pose({"position": [3.0, 2.0, 1.0], "orientation": [0.0, 25.0, 0.0]})][i % 3]
        d_x, d_e = ctx.upload(S), ctx.upload(np.full(n, 0.5 + i))
        ps.append(p); trs.append(t); xs.append(d_x); es.append(d_e); lds_.append(L); min_us.append(0.1 * (i % 4)); ws.append(1.0 + 0.25 * i); als.append(al)
        d_r = ctx.upload(np.full(n, 0.5 + i))
        p.score_trajectory_dev(t, d_x, np.float32, n, L, d_r, min_us[-1], ws[-1], al, accumulate=True)
        ref.append(ctx.download(d_r, (n,), np.float64))
        d_r.free()
    _capi.Primitive.score_trajectories_dev(ps, trs, xs, np.float32, n, lds_, es, min_us, ws, als, accumulate=True)
    for i in range(20):
        got = ctx.download(es[i], (n,), np.float64)
        assert np.array_equal(got.view(np.uint64), ref[i].view(np.uint64)), i
    _capi.Primitive.score_trajectories_dev(ps[:3], trs[:3], xs[:3], np.float32, n, lds_[:3], es[:3], min_us[:3], ws[:3], None, accumulate=False)
    for i in range(3):
        d_r = ctx.malloc(n * 8)
        ps[i].score_trajectory_dev(trs[i], xs[i], np.float32, n, lds_[i], d_r, min_us[i], ws[i], None, accumulate=False)
        assert np.array_equal(ctx.download(es[i], (n,), np.float64), ctx.download(d_r, (n,), np.float64))
        d_r.free()
    for b in xs + es:
        b.free()


def test_per_frame_constraints_through_the_reference_entry_points():
    """evaluate_samples_using_constraints / the sample filter / the objectives take per-frame constraints beside keyframe and
    root-trajectory ones: the errors add up constraint by constraint, the first minimum wins."""
    orc, data, mp, op, joints, animated, sk, S = _setup(n=12, seed=9)
    frames = _constraints(op, orc, S, joints, animated)[:3]
    frames0 = op.back_project_frames(S[0])
    keyframes = [{"type": "position", "t": 100.0, "weight": 1.0, "target": [float(frames0[100, 0]) + 1.0, None, float(frames0[100, 2])]},
                 {"type": "joint_position", "t": 50.0, "weight": 0.5, "target": [10.0, 100.0, 5.0], "joint": "Head"}]
    root_traj = {"type": "trajectory", "control_points": (frames0[::26, :3] + 0.25).tolist(), "min_u": 0.0, "weight": 1.0, "granularity": 1000}
    clist = keyframes + [root_traj] + frames
    ref_frames, _ = op.frame_constraint_errors(S, frames, joints, animated)
    ref_key = op.skeleton_residuals(S, keyframes, joints, animated).sum(axis=1)
    paths = [op.back_project_frames(s)[:, :3] for s in S]
    ref_traj = []
    for path in paths:
        min_u, walk = 0.0, []
        for p in path:
            pt, min_u = orc.closest_point(root_traj["control_points"], p, min_u)
            walk.append(np.linalg.norm(p - pt))
        ref_traj.append(np.mean(walk))
    want = ref_key + np.array(ref_traj) + ref_frames
    got = errors_of_samples(mp._prim, clist, sk, None, S)
    np.testing.assert_allclose(got, want, rtol=2e-6, atol=1e-8)      # (the trajectory searches' stated tolerance)

    class Cons(object):
        constraints = clist
        hip_skeleton = sk
        is_local = True
    best, err = evaluate_samples_using_constraints(S, mp, Cons())
    np.testing.assert_array_equal(best, S[int(np.argmin(want))])
    assert abs(err - want.min()) <= 2e-6 * max(1.0, want.min())
    np.testing.assert_allclose(HipSampleFilter.score_samples(mp, S, Cons(), skeleton=sk), want, rtol=2e-6, atol=1e-8)
    # the optimiser's objectives: the residual vector carries the per-frame blocks behind the fused ones
    from morphablegraphs_amd import objective_functions as of
    np.testing.assert_allclose(of.obj_spatial_error_sum(S[2], (mp, Cons(), None, 1.0, 1.0)), want[2], rtol=2e-6, atol=1e-8)
    r = of.obj_spatial_error_residual_vector(S[2], (mp, Cons(), None, 1.0, 1.0, 2.0))
    fblocks = frame_constraints_errors(mp._prim, S[2:3], frames, sk, None)[1]
    n_frame_entries = sum(b.shape[1] for b in fblocks)
    assert len(r) == len(keyframes) + op.n_canonical_frames + n_frame_entries
    np.testing.assert_allclose(r[-n_frame_entries:], np.concatenate([b[0] for b in fblocks]) / 2.0, rtol=1e-12)


def test_per_frame_constraints_need_a_skeleton_for_other_joints():
    orc, data, mp, op, joints, animated, sk, S = _setup(n=2)
    c = {"type": "frame_ca_position", "joint": "LeftHand", "target": [0.0, 0.0, 0.0], "n_frames": 10, "weight": 1.0}
    with pytest.raises(NotImplementedError):
        frame_constraint_residuals(mp._prim, S, c, None, None)
    # the root alone needs none
    root = {"type": "frame_ca_position", "joint": "root", "target": [80.0, None, 20.0], "n_frames": op.n_canonical_frames, "weight": 1.0}
    res, err = frame_constraint_residuals(mp._prim, S, root, None, None)
    ref = []
    for s in S:
        coeffs = op.back_project_spatial_coeffs(s)
        fr = orc.spline_frames(op.knots, coeffs, np.arange(op.n_canonical_frames, dtype=np.float64))
        ref.append(min(np.hypot(80.0 - f[0], 20.0 - f[2]) for f in fr))
    np.testing.assert_allclose(err, ref, rtol=1e-10)


def test_arc_length_look_up_on_the_device_against_the_reference_vectors():
    """query_point_by_absolute_arc_length on the device (mg_score_frame_constraint's LOCAL_TRAJECTORY branch) against the points the
    reference's own ParameterizedSpline returned (tests/golden/trajectory_spline.npz): a track that climbs straight up in y walks
    exactly the arc lengths of the vectors, so the residual of frame f is the squared xz distance of the looked-up point from the
    y axis."""
    import ctypes as C
    from conftest import load_golden
    from morphablegraphs_amd.candidate_scoring import cached_trajectory
    g = load_golden("trajectory_spline")
    orc, data, mp, op, joints, animated, sk, S = _setup(n=1)
    prim, ctx = mp._prim, mp._prim.ctx
    for ci in range(int(g["n_cases"])):
        order = np.argsort(g["arc_lengths_%d" % ci], kind="stable")
        arcs, want = g["arc_lengths_%d" % ci][order], g["points_by_arc_%d" % ci][order]
        track = np.zeros((1, len(arcs), 1, 3))
        track[0, :, 0, 1] = arcs - arcs[0]
        traj = cached_trajectory(prim, {"control_points": g["control_points_%d" % ci], "granularity": 1000})
        desc = _capi.FrameConstraintDesc()
        desc.type, desc.weight, desc.n_joints, desc.start_arc = _capi.MG_FRAME_LOCAL_TRAJECTORY, 1.0, 1, float(arcs[0])
        desc.trajectories[0] = traj.handle.value
        d_t, d_e, d_r = ctx.upload(track), ctx.malloc(8), ctx.malloc(len(arcs) * 8)
        _capi._check(prim.lib.mg_score_frame_constraint(prim.handle, C.byref(desc), d_t.ptr, 1, len(arcs), 1, d_e.ptr, 0, d_r.ptr))
        res = ctx.download(d_r, (len(arcs),), np.float64)
        np.testing.assert_allclose(res, want[:, 0] ** 2 + want[:, 2] ** 2, rtol=1e-9, atol=1e-9)
        np.testing.assert_allclose(ctx.download(d_e, (1,), np.float64)[0], res.sum(), rtol=1e-12)
        for b in (d_t, d_e, d_r):
            b.free()


def test_planner_step_scores_options_with_per_frame_constraints_like_the_general_chain():
    """evaluate_options_on_device is one launch for keyframe constraints; an option that carries a trajectory or per-frame
    constraint adds those to the errors that launch left on the device and takes its own first minimum (round 3: the whole step
    went option by option) -- the same draws (component counts in option order from NumPy's stream, sampler keyed by seed + option
    index) and the same additions in the same order, so every option's answer is what sample_and_evaluate_on_device gives."""
    from morphablegraphs_amd.candidate_scoring import sample_and_evaluate_on_device
    from morphablegraphs_amd.motion_state_graph import HipPrimitiveSet
    prims = synthetic.make_graph_primitives(3)
    names = [p["name"] for p in prims]
    pset = HipPrimitiveSet(prims)
    cons = {}
    for nm, p in zip(names, prims):
        tl = float(p["n_canonical_frames"] - 1)
        cons[nm] = [{"type": "position", "t": tl, "weight": 1.0, "target": [10.0, None, 5.0]}]
    F1 = prims[1]["n_canonical_frames"]
    cons[names[1]] = cons[names[1]] + [{"type": "frame_ca_position", "joint": "root", "target": [3.0, None, -2.0], "n_frames": F1, "weight": 2.0},
                                       {"type": "trajectory", "control_points": [[0.0, 0.0, 0.0], [5.0, 0.0, 2.0], [12.0, 0.0, 3.0]], "min_u": 0.0, "weight": 0.5,
                                        "granularity": 1000}]
    np.random.seed(17)
    pset.ctx.profile_reset(); pset.ctx.profile_enable(1)
    best, res = pset.evaluate_options_on_device(names, cons, n_samples=777, seed=40)
    # round 4: ONE launch still samples every option and scores its keyframe constraints (no sampler, no scorer launch per option);
    # the options with more add a launch per root trajectory and two per list of per-frame constraints
    assert pset.ctx.profile_get(7)[1] == 1 and pset.ctx.profile_get(4)[1] == 0 and pset.ctx.profile_get(2)[1] == 0
    assert pset.ctx.profile_get("joint_tracks")[1] >= 1 and pset.ctx.profile_get("frame_constraints")[1] >= 1
    pset.ctx.profile_enable(0)
    np.random.seed(17)
    for k, nm in enumerate(names):
        lat, err = sample_and_evaluate_on_device(pset.nodes[nm], cons[nm], 777, seed=40 + k)
        np.testing.assert_array_equal(res[nm][0], lat)
        assert res[nm][1] == err
    assert best == names[int(np.argmin([res[nm][1] for nm in names]))]
    # keyframe-only steps stay one launch
    only = {nm: cons[nm][:1] for nm in names}
    pset.ctx.profile_reset(); pset.ctx.profile_enable(1)
    pset.evaluate_options_on_device(names, only, n_samples=777, seed=40)
    assert pset.ctx.profile_get(7)[1] == 1
    pset.ctx.profile_enable(0)


def test_frame_constraint_entry_points_refuse_bad_arguments():
    """The C-ABI checks what a kernel would otherwise index with: unknown types, a joint count that does not match the tracks, a
    quaternion channel outside the frame, trajectories of another primitive, NULL buffers, a previous-frame alignment without the
    candidates' headings."""
    import ctypes as C
    orc, data, mp, op, joints, animated, sk, S = _setup(n=2)
    prim, ctx, lib = mp._prim, mp._prim.ctx, mp._prim.lib
    other = _primitive(synthetic.make_walk_primitive(seed=1))
    d_t, d_e = ctx.malloc(2 * 4 * 3 * 3 * 8), ctx.malloc(16)

    def call(desc, J=1, tracks=d_t, err=d_e):
        return lib.mg_score_frame_constraint(prim.handle, C.byref(desc), tracks.ptr if tracks is not None else None, 2, 4, J,
                                             err.ptr if err is not None else None, 0, None)
    d = _capi.FrameConstraintDesc()
    d.type, d.weight, d.n_joints = 99, 1.0, 1
    assert call(d) == _capi.MG_ERR_INVALID_ARGUMENT
    d.type = _capi.MG_FRAME_CA_POSITION
    d.axis_on[0], d.target[0] = 1, float("nan")
    assert call(d) == _capi.MG_ERR_INVALID_ARGUMENT                      # a constrained axis without a finite target
    d.target[0] = 1.0
    assert call(d, J=2) == _capi.MG_ERR_INVALID_ARGUMENT                 # one joint's track expected
    assert call(d, tracks=None) == _capi.MG_ERR_INVALID_ARGUMENT
    assert call(d) == 0
    d.type = _capi.MG_FRAME_TRAJECTORY_SET
    d.n_joints = 3
    assert call(d, J=2) == _capi.MG_ERR_INVALID_ARGUMENT                 # n_joints must be the tracks' joint count
    d.n_joints = 2
    assert call(d, J=2) == _capi.MG_ERR_INVALID_ARGUMENT                 # no trajectories
    foreign = _capi.Trajectory(other._prim, np.array([[0.0, 0, 0], [1.0, 0, 0], [2.0, 0, 1.0]]))
    d.trajectories[0] = d.trajectories[1] = foreign.handle.value
    assert call(d, J=2) == _capi.MG_ERR_INVALID_ARGUMENT                 # ... of another primitive
    d.type, d.n_joints, d.quat_channel = _capi.MG_FRAME_JOINT_ROTATION, 1, 2
    d.quaternion[0] = 1.0
    assert call(d, J=9) == _capi.MG_ERR_INVALID_ARGUMENT                 # channel 2 is root translation
    d.quat_channel = 7
    assert call(d, J=9) == _capi.MG_ERR_INVALID_ARGUMENT                 # 7 + 4 > 9
    d.quaternion[0] = 0.0
    d.quat_channel = 3
    assert call(d, J=9) == _capi.MG_ERR_INVALID_ARGUMENT                 # a zero rotation target
    # mg_align_frames: a previous-frame record needs the candidates' headings
    al = _capi.ConstraintSet._marshal_alignment({"joint": 0, "position": (0.0, 0.0, 0.0), "heading": (0.0, 1.0)}, None)
    d_f, d_v = ctx.malloc(2 * 4 * prim.n_dim * 8), ctx.malloc(2 * 4 * 8)
    assert lib.mg_align_frames(prim.handle, d_f.ptr, 2, 4, d_v.ptr, 2, C.byref(al)) == _capi.MG_ERR_INVALID_ARGUMENT
    assert lib.mg_align_frames(prim.handle, d_f.ptr, 2, 4, d_v.ptr, 3, C.byref(al)) == _capi.MG_ERR_INVALID_ARGUMENT
    al.heading[0], al.heading[1] = 0.0, 0.0
    assert lib.mg_align_frames(prim.handle, d_f.ptr, 2, 4, d_v.ptr, 4, C.byref(al)) == _capi.MG_ERR_INVALID_ARGUMENT
    # round 4's entry points: the track plan, the tracks, the list scorer, the scorers side by side
    vp = C.c_void_p
    skd = sk.desc()
    h = vp()
    one = (C.c_int32 * 1)(1)
    hand = (C.c_int32 * 1)(sk.index("LeftHand"))
    assert lib.mg_track_plan_create(prim.handle, C.byref(skd), 0, one, hand, 0, C.byref(h)) == _capi.MG_ERR_INVALID_ARGUMENT        # no request
    assert lib.mg_track_plan_create(prim.handle, C.byref(skd), 5, (C.c_int32 * 5)(1, 1, 1, 1, 1), (C.c_int32 * 5)(1, 2, 3, 4, 5), 0, C.byref(h)) == _capi.MG_ERR_INVALID_ARGUMENT
    too_many = _capi.MG_FRAME_MAX_JOINTS + 1
    assert lib.mg_track_plan_create(prim.handle, C.byref(skd), 1, (C.c_int32 * 1)(too_many), (C.c_int32 * too_many)(*range(1, too_many + 1)), 0, C.byref(h)) == _capi.MG_ERR_INVALID_ARGUMENT   # too many joints in a request
    assert lib.mg_track_plan_create(prim.handle, C.byref(skd), 1, one, (C.c_int32 * 1)(999), 0, C.byref(h)) == _capi.MG_ERR_INVALID_ARGUMENT   # no such joint
    assert lib.mg_track_plan_create(prim.handle, C.byref(skd), 1, one, hand, 999, C.byref(h)) == _capi.MG_ERR_INVALID_ARGUMENT                # no such aligning joint
    plan = _capi.TrackPlan(prim, sk, [["LeftHand"]], align_joint=0)
    d_S, d_o = ctx.upload(S.astype(np.float32)), ctx.malloc(2 * prim.n_canonical_frames * 3 * 8)
    grids, outs = (vp * 1)(None), (vp * 1)(d_o.ptr.value)
    call_tracks = lambda lat, dt, n, ld, al=None, g=grids, o=outs: lib.mg_joint_tracks(plan.handle, lat, dt, n, ld, al, g, o)
    assert call_tracks(d_S.ptr, _capi.MG_F32, 2, 40) == 0
    assert call_tracks(None, _capi.MG_F32, 2, 40) == _capi.MG_ERR_INVALID_ARGUMENT
    assert call_tracks(d_S.ptr, _capi.MG_F32, 2, 39) == _capi.MG_ERR_INVALID_ARGUMENT            # ld < n_components
    assert call_tracks(d_S.ptr, 7, 2, 40) == _capi.MG_ERR_INVALID_ARGUMENT                       # no such dtype
    assert call_tracks(d_S.ptr, _capi.MG_F32, 2, 40, o=(vp * 1)(None)) == _capi.MG_ERR_INVALID_ARGUMENT
    foreign_grid = other._prim.time_grid(np.arange(4.0))
    assert call_tracks(d_S.ptr, _capi.MG_F32, 2, 40, g=(vp * 1)(foreign_grid.handle)) == _capi.MG_ERR_INVALID_ARGUMENT
    al_other = _capi.ConstraintSet._marshal_alignment({"joint": sk.index("Spine1"), "position": (0.0, 0.0, 0.0), "heading": (0.0, 1.0)}, sk)
    assert call_tracks(d_S.ptr, _capi.MG_F32, 2, 40, al=C.byref(al_other)) == _capi.MG_ERR_INVALID_ARGUMENT   # the plan aligns through the root
    al_zero = _capi.ConstraintSet._marshal_alignment({"joint": 0, "position": (0.0, 0.0, 0.0), "heading": (0.0, 0.0)}, sk)
    assert call_tracks(d_S.ptr, _capi.MG_F32, 2, 40, al=C.byref(al_zero)) == _capi.MG_ERR_INVALID_ARGUMENT
    dd = _capi.FrameConstraintDesc()
    dd.type, dd.weight, dd.n_joints = _capi.MG_FRAME_CA_POSITION, 1.0, 1
    dd.axis_on[0], dd.target[0] = 1, 1.0
    dptr, tptr = (vp * 1)(C.addressof(dd)), (vp * 1)(d_o.ptr.value)
    tT, tJ = (C.c_int32 * 1)(prim.n_canonical_frames), (C.c_int32 * 1)(1)
    assert lib.mg_score_frame_constraints(prim.handle, 1, dptr, tptr, tT, tJ, 2, d_e.ptr, 0, None) == 0
    assert lib.mg_score_frame_constraints(prim.handle, 1, dptr, (vp * 1)(None), tT, tJ, 2, d_e.ptr, 0, None) == _capi.MG_ERR_INVALID_ARGUMENT
    assert lib.mg_score_frame_constraints(prim.handle, 1, dptr, tptr, tT, tJ, 2, None, 0, None) == _capi.MG_ERR_INVALID_ARGUMENT
    assert lib.mg_score_frame_constraints(prim.handle, 1, dptr, tptr, tT, (C.c_int32 * 1)(2), 2, d_e.ptr, 0, None) == _capi.MG_ERR_INVALID_ARGUMENT
    assert lib.mg_score_frame_constraints(prim.handle, -1, dptr, tptr, tT, tJ, 2, d_e.ptr, 0, None) == _capi.MG_ERR_INVALID_ARGUMENT
    traj = _capi.Trajectory(prim, np.array([[0.0, 0, 0], [1.0, 0, 0], [2.0, 0, 1.0]]))
    two = lambda prims_, trajs_: lib.mg_score_trajectories(2, (vp * 2)(*prims_), (vp * 2)(*trajs_), (vp * 2)(d_S.ptr.value, d_S.ptr.value), _capi.MG_F32, 2,
                                                       (C.c_int64 * 2)(40, 40), (C.c_double * 2)(0.0, 0.0), (C.c_double * 2)(1.0, 1.0), None,
                                                       (vp * 2)(d_e.ptr.value, d_e.ptr.value), 0)
    assert two([prim.handle.value] * 2, [traj.handle.value] * 2) == 0
    assert two([prim.handle.value] * 2, [traj.handle.value, foreign.handle.value]) == _capi.MG_ERR_INVALID_ARGUMENT      # a trajectory of another primitive
    assert two([prim.handle.value, None], [traj.handle.value] * 2) == _capi.MG_ERR_INVALID_ARGUMENT
    ctx.synchronize()
    plan.close()
    traj.close()
    foreign_grid.close()
    foreign.close()
    for b in (d_t, d_e, d_f, d_v, d_S, d_o):
        b.free()


def test_cached_track_scorers_keep_their_trajectories_alive():
    """a TrackScorer's records carry raw mg_trajectory handles; the trajectory cache holds a bounded number of
    entries and closes what it evicts.  More distinct (primitive, trajectory) pairs than the cache holds, across scorers that stay
    alive: every scorer still scores what it scored when it was new (its trajectories are pinned), and once the scorers are closed
    the evicted trajectories go."""
    from morphablegraphs_amd import candidate_scoring as cs, frame_constraints as fc
    orc, data, mp, op, joints, animated, sk, S = _setup(n=17, seed=8)
    prim, ctx = mp._prim, mp._prim.ctx
    frames0 = op.back_project_frames(S[0])
    hand = np.array([orc.joint_global_position(f, joints, animated, "LeftHand") for f in frames0])
    n_scorers = cs._TRAJ_CACHE_SIZE + 9
    lists = [[{"type": "frame_joint_trajectory", "joint": "LeftHand", "control_points": (hand[::22] + np.array([0.1 * k, 0.0, -0.05 * k])).tolist(),
               "min_u": 0.0, "granularity": 1000, "weight": 1.0},
              {"type": "frame_local_trajectory", "joint": "Hips", "control_points": (frames0[::20, :3] + np.array([0.02 * k, 0.0, 0.5])).tolist(),
               "granularity": 1000, "start_t": 1.0, "n_frames": op.n_canonical_frames, "weight": 0.5}] for k in range(n_scorers)]
    lat = S.astype(np.float32)
    d_S, d_e = ctx.upload(lat), ctx.malloc(len(S) * 8)
    scorers, first = [], []
    try:
        for cl in lists:
            sc = fc.TrackScorer(prim, cl, sk, None)
            sc.score_dev(d_S, lat.dtype, len(S), lat.shape[1], d_e, accumulate=False)
            first.append(ctx.download(d_e, (len(S),), np.float64))
            scorers.append(sc)
        assert len(cs._TRAJ_CACHE) <= cs._TRAJ_CACHE_SIZE
        kept = [t for sc in scorers for t in sc.keep]
        assert all(t.handle for t in kept) and sum(1 for t in kept if t.evicted) >= 2 * 9
        for sc, want in zip(scorers, first):       # every scorer again, the oldest ones long after the cache dropped their entries
            assert sc.valid()
            sc.score_dev(d_S, lat.dtype, len(S), lat.shape[1], d_e, accumulate=False)
            np.testing.assert_array_equal(ctx.download(d_e, (len(S),), np.float64).view(np.uint64), want.view(np.uint64))
        # and they are what the chain gives
        total = frame_constraints_errors(prim, lat, lists[0], sk, None)[0]
        np.testing.assert_array_equal(total.view(np.uint64), first[0].view(np.uint64))
    finally:
        for sc in scorers:
            sc.close()
        d_S.free(); d_e.free()
    assert all((not t.handle) for t in kept if t.evicted)          # the last holder closed what the cache had let go of
    # a cleared cache under a live scorer: the scorer says so instead of launching on dangling pointers
    sc = fc.TrackScorer(prim, lists[0], sk, None)
    cs.clear_constraint_cache()
    assert sc.valid()           # pinned: still alive
    sc.close()


def test_mixed_lists_add_in_list_order():
    """A joint-rotation constraint BETWEEN track constraints: the additions happen in the list's order (two fused runs around the
    rotation's chain launch), so the sum has the bits of the chain that adds one constraint after the other."""
    from morphablegraphs_amd import frame_constraints as fc
    orc, data, mp, op, joints, animated, sk, S = _setup(n=21, seed=5)
    cl = _constraints(op, orc, S, joints, animated)
    mixed = [cl[1], cl[5], cl[0], cl[2], cl[6], cl[3]]
    total, blocks = frame_constraints_errors(mp._prim, S, mixed, sk, None)
    fc.FUSED = False
    try:
        total_c, blocks_c = frame_constraints_errors(mp._prim, S, mixed, sk, None)
    finally:
        fc.FUSED = True
    for b, bc in zip(blocks, blocks_c):
        assert np.array_equal(b.view(np.uint64), bc.view(np.uint64))
    assert np.array_equal(total.view(np.uint64), total_c.view(np.uint64))


def test_planner_step_with_per_frame_lists_is_four_launches():
    """A planner step whose options carry root trajectories AND per-frame lists: one launch draws and keyframe-
    scores every option (mg_options_step), one scores the options' trajectories side by side (mg_score_trajectories), one makes every
    option's joint tracks and one adds every option's list and takes every option's first minimum (mg_options_frame_lists) -- four
    launches whatever the number of options, the winners those of the per-option general chain (sample_and_evaluate_on_device), bit
    for bit; an option the fast form does not take (a joint-rotation constraint in its list) goes the per-option way in the same step."""
    from morphablegraphs_amd.candidate_scoring import sample_and_evaluate_on_device
    from morphablegraphs_amd.motion_state_graph import HipPrimitiveSet
    prims = synthetic.make_graph_primitives(6)
    names = [p["name"] for p in prims]
    joints, animated = synthetic.make_skeleton()
    sk = _capi.Skeleton(joints, animated)
    pset = HipPrimitiveSet(prims)
    cons = {}
    for i, (nm, p) in enumerate(zip(names, prims)):
        F = p["n_canonical_frames"]
        clist = [{"type": "position", "t": float(F - 1), "weight": 1.0, "target": [10.0, None, 5.0]}]
        if i != 1:
            clist.append({"type": "trajectory", "control_points": [[0.0, 0.0, 0.0], [5.0 + i, 0.0, 2.0], [12.0, 0.0, 3.0 + i]], "min_u": 0.0, "weight": 0.5, "granularity": 1000})
        if i in (0, 2, 3):
            clist.append({"type": "frame_ca_position", "joint": "LeftHand", "target": [3.0 + i, None, -2.0], "n_frames": F, "weight": 2.0})
        if i == 3:
            clist.append({"type": "frame_discrete_trajectory", "joint": "RightFoot", "points": [[0.1 * f, 5.0, 0.2 * f] for f in range(F // 2)], "unconstrained": [1], "weight": 0.4})
            clist.append({"type": "frame_joint_trajectory", "joint": "LeftHand", "control_points": [[0.0, 90.0, 0.0], [20.0, 95.0, 5.0], [40.0, 90.0, 12.0]], "min_u": 0.0,
                          "granularity": 1000, "weight": 0.3})
        if i == 4:
            clist.append({"type": "frame_joint_rotation", "joint_index": 3, "quaternion": [0.9, 0.1, -0.3, 0.2], "frame_idx": 7.0, "weight": 0.6})
        cons[nm] = clist

    class Cons(object):
        def __init__(self, cl):
            self.constraints, self.hip_skeleton, self.is_local = cl, sk, True
    wrapped = {nm: Cons(cons[nm]) for nm in names}
    for n_samples in (777, 4096):
        np.random.seed(17)
        pset.ctx.profile_reset(); pset.ctx.profile_enable(1)
        best, res = pset.evaluate_options_on_device(names, wrapped, n_samples=n_samples, seed=40)
        counts = {slot: pset.ctx.profile_get(slot)[1] for slot in ("options_step", "trajectory", "joint_tracks", "frame_constraints")}
        pset.ctx.profile_enable(0)
        # option 4's joint-rotation constraint takes the frames chain (its own launches); everything else: one launch per kind
        assert counts["options_step"] == 1 and counts["trajectory"] == 1 and counts["joint_tracks"] == 1, counts
        assert counts["frame_constraints"] <= 2, counts
        np.random.seed(17)
        for k, nm in enumerate(names):
            lat, err = sample_and_evaluate_on_device(pset.nodes[nm], wrapped[nm], n_samples, seed=40 + k)
            np.testing.assert_array_equal(res[nm][0], lat, err_msg=nm)
            assert res[nm][1] == err, (nm, res[nm][1], err)
        assert best == names[int(np.argmin([res[nm][1] for nm in names]))]
