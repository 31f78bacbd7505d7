"""GPU tests of the trajectory constraints' closest-point search against vectors the REFERENCE's own function produced
(tests/golden/trajectory_closest_point.npz: ParameterizedSpline.find_closest_point_fast, parameterized_spline.py:303-322, chained
frame to frame as trajectory_constraint.py:93-116 chains it; oracle/gen_golden.py run_closest_point_case), and of the per-frame
classes downstream of forward kinematics against the reference's own classes (tests/golden/per_frame_classes.npz:
LocalTrajectoryConstraint, TrajectorySetConstraint with a duck-typed skeleton that returns given joint tracks).

TOLERANCE of the search, two-sided, stated: |u - u_ref| <= 2e-6 of the parameter range and |d - d_ref| <= 1e-6 max(1, d_ref) per
frame, over whole chains of 156 frames (a frame's bound is the previous frame's result, so drift would accumulate).  What sets it:
the reference's gradient is a forward difference with h = 1e-8, which amplifies the last bit of the distance by 1e8 -- two correct
implementations of the same algorithm agree to ~1e-8 per search (the reference's scipy 1.15 and scipy 1.7 differ from each other at
that level), and L-BFGS-B stops at ftol = 2.2e-9 / gtol = 1e-5, not at the minimum.  The test prints the measured deviations."""
import ctypes as C
import os

import numpy as np
import pytest

from morphablegraphs_amd import _capi
from test_gpu_adaptors import _path_following_model, _primitive

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
U_TOL, D_TOL = 2.0e-6, 1.0e-6


def _cases():
    g = np.load(os.path.join(GOLDEN, "trajectory_closest_point.npz"))
    for ci in range(int(g["n_cases"])):
        cps = g["control_points_%d" % ci]
        for ti, name in enumerate(g["track_names_%d" % ci]):
            yield ci, str(name), cps, g["track_%d_%d" % (ci, ti)], float(g["min_u0_%d_%d" % (ci, ti)]), g["u_%d_%d" % (ci, ti)], \
                g["distance_%d_%d" % (ci, ti)], g["target_%d_%d" % (ci, ti)]


def test_closest_point_search_is_the_references_search():
    """mg_trajectory_closest_points (the search alone), mg_score_trajectory_points (TrajectoryConstraint's residual vector and
    error) and the scorer's MG_FRAME_JOINT_TRAJECTORY type on the golden tracks: 2-, 5-, 9- and 7-point splines; tracks near, far,
    beyond the end, with a late start, running backwards, a hand off the plane, the root paths of the golden walks."""
    from oracle import mg_oracle as orc
    mp = _primitive(_path_following_model())
    prim, ctx = mp._prim, mp._prim.ctx
    worst_u, worst_d, report = 0.0, 0.0, []
    for ci, name, cps, track, min_u0, u_ref, d_ref, target_ref in _cases():
        traj = _capi.Trajectory(prim, cps, granularity=1000)
        try:
            u, d = prim.trajectory_closest_points(traj, track[None], min_u0)
            du, dd = np.abs(u[0] - u_ref), np.abs(d[0] - d_ref) / np.maximum(1.0, d_ref)
            report.append("case %d %-10s max|du| %.2e  max|dd|/max(1,d) %.2e  drift at the last frame %.2e" % (ci, name, du.max(), dd.max(), du[-1]))
            worst_u, worst_d = max(worst_u, du.max()), max(worst_d, dd.max())
            assert du.max() <= U_TOL and dd.max() <= D_TOL, report[-1]
            # the points themselves
            pts = np.array([orc.catmull_rom_point(cps, x) for x in u[0]])
            assert np.abs(pts - target_ref).max() <= 1e-5 * max(1.0, np.abs(target_ref).max())
            # the constraint's numbers: residual vector = weight * distance, error = its average
            err, res = prim.score_trajectory_points(traj, track[None], min_u0, weight=1.5, residuals=True)
            np.testing.assert_array_equal(res[0], 1.5 * d[0])
            assert abs(err[0] - 1.5 * d_ref.mean()) <= 1.5 * D_TOL * max(1.0, d_ref.max())
            # the same constraint as a member of the scorer's list
            desc = _capi.FrameConstraintDesc()
            desc.type, desc.weight, desc.n_joints, desc.start_arc = _capi.MG_FRAME_JOINT_TRAJECTORY, 1.5, 1, min_u0
            desc.trajectories[0] = traj.handle.value
            n = 70                                                     # (more than one wave: the list kernel, not the points launch's)
            d_t, d_e, d_r = ctx.upload(np.repeat(track[None], n, axis=0)), ctx.malloc(n * 8), ctx.malloc(n * len(track) * 8)
            _capi._check(prim.lib.mg_score_frame_constraint(prim.handle, C.byref(desc), d_t.ptr, n, len(track), 1, d_e.ptr, 0, d_r.ptr))
            res_l = ctx.download(d_r, (n, len(track)), np.float64)
            for b in (d_t, d_e, d_r):
                b.free()
            np.testing.assert_array_equal(res_l, np.repeat(res, n, axis=0))
        finally:
            traj.close()
    print("\n".join(report))
    print("worst |du| %.3e (tolerance %.1e), worst relative |dd| %.3e (tolerance %.1e)" % (worst_u, U_TOL, worst_d, D_TOL))


def test_device_search_is_the_oracles_restatement():
    """... and against oracle/mg_oracle.py closest_point_lbfgsb (the restatement the device mirrors, formk's verdict by the device's
    rule) on tracks that are NOT in the fixture: the same parameters to the forward difference's noise."""
    from oracle import mg_oracle as orc
    mp = _primitive(_path_following_model())
    prim = mp._prim
    rng = np.random.default_rng(12)
    for n_points in (3, 6, 11):
        cps = np.cumsum(np.column_stack([rng.uniform(10, 50, n_points), rng.uniform(-4, 4, n_points), rng.uniform(-30, 30, n_points)]), axis=0)
        s = np.linspace(0.0, 1.0, 48)
        tracks = np.stack([cps[0] + (lo + (hi - lo) * s)[:, None] * (cps[-1] - cps[0]) + rng.normal(0.0, amp, (48, 3))
                           for lo, hi, amp in ((0.0, 1.0, 1.0), (0.2, 0.8, 8.0), (0.0, 1.3, 0.3), (0.5, 0.1, 2.0))])
        traj = _capi.Trajectory(prim, cps, granularity=1000)
        u, d = prim.trajectory_closest_points(traj, tracks, 0.05)
        traj.close()
        for b, track in enumerate(tracks):
            mu, us = 0.05, []
            for p in track:
                _, mu = orc.closest_point_lbfgsb(cps, p, mu, formk_rule=True)
                us.append(mu)
            assert np.abs(u[b] - np.array(us)).max() <= U_TOL, (n_points, b, np.abs(u[b] - np.array(us)).max())


def test_monotone_walk_stays_available_and_differs_where_the_distance_has_several_basins():
    """MG_OPT_TRAJECTORY_SEARCH 1: the deterministic walk of rounds 2-4, held to its own restatement (closest_point_walk, 1e-9) -- and
    the measured reason it is not the default: on the fixture's multi-basin tracks it ends far from the reference."""
    from oracle import mg_oracle as orc
    mp = _primitive(_path_following_model())
    prim, ctx = mp._prim, mp._prim.ctx
    ctx.set_option(_capi.MG_OPT_TRAJECTORY_SEARCH, 1)
    try:
        far_off = 0
        for ci, name, cps, track, min_u0, u_ref, d_ref, _ in _cases():
            if ci not in (1, 3):
                continue
            traj = _capi.Trajectory(prim, cps, granularity=1000)
            u, d = prim.trajectory_closest_points(traj, track[None], min_u0)
            traj.close()
            mu, us = min_u0, []
            for p in track[:40]:
                _, mu = orc.closest_point_walk(cps, p, mu)
                us.append(mu)
            np.testing.assert_allclose(u[0, :40], us, rtol=0, atol=1e-8)
            far_off += int(np.abs(u[0] - u_ref).max() > 0.1)
        assert far_off >= 4
    finally:
        ctx.set_option(_capi.MG_OPT_TRAJECTORY_SEARCH, 0)


def test_local_trajectory_and_trajectory_set_against_the_references_classes():
    """mg_score_frame_constraint on GIVEN joint tracks against LocalTrajectoryConstraint.get_residual_vector_spline /
    evaluate_motion_spline (local_trajectory_constraint.py:45-78) and TrajectorySetConstraint.get_residual_vector /
    evaluate_motion_sample (trajectory_set_constraint.py:41-104) run by the reference itself: 1e-9 relative."""
    g = np.load(os.path.join(GOLDEN, "per_frame_classes.npz"))
    mp = _primitive(_path_following_model())
    prim, ctx = mp._prim, mp._prim.ctx
    lib = prim.lib

    def score(desc, tracks, J):
        tracks = np.ascontiguousarray(tracks, dtype=np.float64)       # (T, J, 3)
        T = tracks.shape[0]
        m = lib.mg_frame_constraint_width(C.byref(desc), T)
        d_t, d_e, d_r = ctx.upload(tracks[None]), ctx.malloc(8), ctx.malloc(max(m, 1) * 8)
        try:
            _capi._check(lib.mg_score_frame_constraint(prim.handle, C.byref(desc), d_t.ptr, 1, T, J, d_e.ptr, 0, d_r.ptr))
            return ctx.download(d_r, (m,), np.float64), float(ctx.download(d_e, (1,), np.float64)[0])
        finally:
            for b in (d_t, d_e, d_r):
                b.free()
    for ci in range(int(g["n_cases"])):
        cps, cps2, hips, hand = g["control_points_%d" % ci], g["control_points2_%d" % ci], g["hips_%d" % ci], g["hand_%d" % ci]
        t1, t2 = _capi.Trajectory(prim, cps, 1000), _capi.Trajectory(prim, cps2, 1000)
        try:
            for si in range(2):
                nf = int(g["local_n_frames_%d_%d" % (ci, si)])
                desc = _capi.FrameConstraintDesc()
                desc.type, desc.weight, desc.n_joints, desc.n_frames = _capi.MG_FRAME_LOCAL_TRAJECTORY, 1.0, 1, nf
                desc.start_arc = float(g["local_start_t_%d_%d" % (ci, si)])
                desc.trajectories[0] = t1.handle.value
                res, err = score(desc, hips[:nf, None, :], 1)
                want = g["local_residuals_%d_%d" % (ci, si)]
                np.testing.assert_allclose(res, want, rtol=1e-9, atol=1e-9 * max(1.0, want.max()))
                assert abs(err - float(g["local_error_%d_%d" % (ci, si)])) <= 1e-9 * max(1.0, abs(float(g["local_error_%d_%d" % (ci, si)])))
            for si in range(3):
                ranges, arcs = g["set_ranges_%d_%d" % (ci, si)], g["set_arc_lengths_%d_%d" % (ci, si)]
                desc = _capi.FrameConstraintDesc()
                desc.type, desc.weight, desc.n_joints, desc.n_frames = _capi.MG_FRAME_TRAJECTORY_SET, 1.0, 2, len(hips)
                for j, t in enumerate((t1, t2)):
                    desc.trajectories[j] = t.handle.value
                    desc.arc0[j] = float(arcs[j])
                    desc.has_range[j] = 0 if np.isnan(ranges[j, 0]) else 1
                    desc.range_start[j], desc.range_end[j] = (0.0, 0.0) if np.isnan(ranges[j, 0]) else (float(ranges[j, 0]), float(ranges[j, 1]))
                res, err = score(desc, np.stack([hips, hand], axis=1), 2)
                want = g["set_residuals_%d_%d" % (ci, si)]
                np.testing.assert_allclose(res, want, rtol=1e-9, atol=1e-9 * max(1.0, want.max()))
                assert abs(err - float(g["set_error_%d_%d" % (ci, si)])) <= 1e-9 * max(1.0, abs(float(g["set_error_%d_%d" % (ci, si)])))
        finally:
            t1.close()
            t2.close()
