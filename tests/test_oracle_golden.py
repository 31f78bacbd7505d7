"""The NumPy oracle against the golden vectors made by the reference's own code
(oracle/gen_golden.py).  CPU only."""
import os

import numpy as np
import pytest

from oracle import mg_oracle as orc


def test_back_project_coeffs_and_frames(golden_case):
    name, data, g = golden_case
    prim = orc.OraclePrimitive(data)
    S = g["S"]
    for b in range(S.shape[0]):
        c = prim.back_project_spatial_coeffs(S[b])
        np.testing.assert_allclose(c, g["coeffs"][b], rtol=0, atol=1e-12 * max(1.0, np.abs(g["coeffs"][b]).max()))
        f = prim.back_project_frames(S[b])
        assert f.shape == g["frames"][b].shape                      # (F, D) indexing is exact
        np.testing.assert_allclose(f, g["frames"][b], rtol=0, atol=2e-12 * max(1.0, np.abs(g["frames"][b]).max()))
    fb = prim.back_project_frames_batch(S)
    np.testing.assert_allclose(fb, g["frames"], rtol=0, atol=2e-12 * max(1.0, np.abs(g["frames"]).max()))


def test_time_function_and_knots(golden_case):
    name, data, g = golden_case
    prim = orc.OraclePrimitive(data)
    np.testing.assert_array_equal(prim.canonical_time_function(), g["time_function"])
    np.testing.assert_array_equal(orc.cubic_b_spline_knots(prim.n_basis, prim.n_canonical_frames), g["knots"])
    assert prim.n_components == int(g["n_spatial_components"])


def test_evaluate_arbitrary_times(golden_case):
    name, data, g = golden_case
    prim = orc.OraclePrimitive(data)
    S, times = g["S"], g["eval_times"]
    for b in range(S.shape[0]):
        c = prim.back_project_spatial_coeffs(S[b])
        e = orc.spline_frames(prim.knots, c, times)
        np.testing.assert_allclose(e, g["evals"][b], rtol=0, atol=2e-12 * max(1.0, np.abs(g["evals"][b]).max()))
    c0 = prim.back_project_spatial_coeffs(S[0])
    for i, t in enumerate(times):
        e = orc.spline_frames(prim.knots, c0, [t])[0]
        np.testing.assert_allclose(e, g["evals_scalar"][i], rtol=0, atol=2e-12 * max(1.0, np.abs(g["evals_scalar"]).max()))


def test_splev_against_installed_scipy():
    import scipy.interpolate as si
    rng = np.random.default_rng(3)
    for nb, F in ((7, 12), (31, 156), (8, 40), (4, 9)):
        knots = orc.cubic_b_spline_knots(nb, F)
        c = rng.standard_normal(nb)
        x = np.concatenate([np.linspace(-1.0, F + 1.0, 301), knots])
        np.testing.assert_allclose(orc.splev(x, knots, c), si.splev(x, (knots, c, 3)), rtol=0, atol=1e-12)


def test_precision_cholesky_and_log_prob(golden_case):
    name, data, g = golden_case
    prim = orc.OraclePrimitive(data)
    pc = g["precisions_cholesky"]
    np.testing.assert_allclose(prim.prec_chol, pc, rtol=1e-9, atol=1e-9 * np.abs(pc).max())
    lp = prim.score_samples(g["X"])
    np.testing.assert_allclose(lp, g["logp"], rtol=1e-9, atol=1e-7)
    np.testing.assert_allclose(prim.score_samples(g["S"]), g["logp_S"], rtol=1e-9, atol=1e-7)
    np.testing.assert_allclose(lp.mean(), float(g["score_mean"]), rtol=1e-9, atol=1e-7)


def test_sample_matches_sklearn_stream(golden_case):
    name, data, g = golden_case
    prim = orc.OraclePrimitive(data)
    np.random.seed(int(g["seed"]))
    S = prim.sample_low_dimensional_vector(g["S"].shape[0])
    np.testing.assert_allclose(S, g["S"], rtol=1e-12, atol=1e-12)


def test_first_min_argmin_rule():
    assert orc.first_min_argmin([3.0, 1.0, 1.0, 2.0]) == (1, 1.0)
    assert orc.first_min_argmin([]) == (0, np.inf)
    assert orc.first_min_argmin([np.nan, 2.0, np.nan, 2.0]) == (1, 2.0)
    assert orc.first_min_argmin([np.inf, np.inf])[0] == 0


def test_point_distance_ignores_none_axes():
    assert orc.point_distance([1.0, None, 3.0], [0.0, 100.0, 1.0]) == np.sqrt(1.0 + 4.0)
    assert orc.point_distance([None, None, None], [1.0, 2.0, 3.0]) == 0.0


def test_log_likelihood_jac_is_minus_the_gradient_of_the_pinned_log_prob(golden_case):
    """objective_functions.py:95-107 restated in the oracle.  The reference module cannot be imported (anim_utils),
    so the restatement is pinned through the identity jac(s) == -d/ds log p(s), with log p pinned by the goldens
    (test_precision_cholesky_and_log_prob): central differences of the oracle's own score_samples."""
    name, data, g = golden_case
    prim = orc.OraclePrimitive(data)
    X = np.vstack([np.asarray(g["S"], dtype=np.float64), np.asarray(g["X"][:4], dtype=np.float64)])
    jac = prim.log_likelihood_jac(X)
    assert jac.shape == X.shape
    scale = np.sqrt(np.array([np.diag(c) for c in prim.covars]).mean(axis=0))   # per-dimension step scale
    checked = 0
    for b in range(X.shape[0]):
        if np.exp(orc.gmm_log_prob(X[b][None], prim.weights, prim.means, prim.prec_chol)[0]) == 0.0:
            np.testing.assert_array_equal(jac[b], np.ones(X.shape[1]))   # the reference's underflow branch
            continue
        checked += 1
        for i in range(X.shape[1]):
            h = 1e-5 * scale[i]
            xp, xm = X[b].copy(), X[b].copy()
            xp[i] += h
            xm[i] -= h
            fd = -(orc.gmm_log_prob(xp[None], prim.weights, prim.means, prim.prec_chol)[0]
                   - orc.gmm_log_prob(xm[None], prim.weights, prim.means, prim.prec_chol)[0]) / (2 * h)
            assert abs(fd - jac[b, i]) <= 2e-6 * max(1.0, abs(jac[b, i]), abs(fd)), (name, b, i, fd, jac[b, i])
    assert checked >= 2
    # denominator underflow: exp(score) == 0 -> ones, like the reference
    far = prim.means[0] + 1e4 * scale
    np.testing.assert_array_equal(prim.log_likelihood_jac(far[None])[0], np.ones(X.shape[1]))


def test_keyframe_residuals_sum_to_the_evaluate_error():
    """get_residual_vector (motion_primitive_constraints.py:124-144) vs evaluate (:100-122): same terms."""
    from morphablegraphs_amd import synthetic
    data = synthetic.make_tiny_primitive(seed=3)
    prim = orc.OraclePrimitive(data)
    rng = np.random.default_rng(0)
    S = rng.standard_normal((5, prim.n_components))
    t = float(prim.n_canonical_frames - 1)
    cons = [{"type": "position", "t": t, "weight": 2.0, "target": [1.0, None, -2.0]},
            {"type": "direction", "t": 0.5 * t, "weight": 0.5, "target": [0.3, 1.0]}]
    res = prim.keyframe_residuals(S, cons)
    assert res.shape == (5, 2)
    np.testing.assert_allclose(res.sum(axis=1), prim.keyframe_errors(S, cons), rtol=1e-14)


def test_forward_kinematics_oracle_known_answers():
    """The self-defined FK oracle (anim_utils is absent): identity pose = sum of offsets along the chain;
    a quarter turn of the root about y maps (x, z) -> (z, -x); rotating a mid-chain joint moves only what hangs
    below it; quaternions need not be unit (normalised like transformations.quaternion_matrix)."""
    from morphablegraphs_amd import synthetic
    joints, animated = synthetic.make_skeleton()
    frame = np.zeros(3 + 4 * len(animated))
    frame[3::4] = 1.0
    frame[:3] = [1.0, 2.0, 3.0]
    np.testing.assert_allclose(orc.joint_global_position(frame, joints, animated, "LeftHand_EndSite"), [76.0, 35.0, 3.5], atol=1e-12)
    np.testing.assert_allclose(orc.joint_global_position(frame, joints, animated, "Hips"), [1.0, 2.0, 3.0], atol=0)
    f2 = frame.copy()
    f2[3:7] = [3.0, 0.0, 3.0, 0.0]                                   # unnormalised quarter turn about y
    np.testing.assert_allclose(orc.joint_global_position(f2, joints, animated, "LeftHand_EndSite"), [1.5, 35.0, -72.0], atol=1e-12)
    f3 = frame.copy()
    ch = 3 + 4 * animated.index("LeftArm")
    f3[ch:ch + 4] = [np.sqrt(0.5), 0.0, 0.0, np.sqrt(0.5)]           # left arm a quarter turn about z: x -> y
    np.testing.assert_allclose(orc.joint_global_position(f3, joints, animated, "LeftArm"),
                               orc.joint_global_position(frame, joints, animated, "LeftArm"), atol=1e-12)
    base = orc.joint_global_position(frame, joints, animated, "LeftArm")
    np.testing.assert_allclose(orc.joint_global_position(f3, joints, animated, "LeftHand_EndSite"), base + [0.0, 60.0, 0.0], atol=1e-12)


def test_alignment_oracle_known_answers():
    """The self-defined 2-D alignment oracle (anim_utils is absent): after aligning, the first control point's root
    sits on the previous root in x and z (y untouched), the aligning node's heading in the first control point is
    the previous heading, control points move rigidly, a motion that already continues the previous one is left
    alone, and a quarter-turn case comes out as computed by hand."""
    from morphablegraphs_amd import synthetic
    joints, animated = synthetic.make_skeleton()
    prim = orc.OraclePrimitive(synthetic.make_walk_primitive(seed=0))
    rng = np.random.default_rng(3)
    coeffs = prim.back_project_spatial_coeffs(rng.standard_normal(prim.n_components))
    prev = prim.back_project_frames(rng.standard_normal(prim.n_components))[-1].copy()
    prev[:3] = [40.0, 95.0, -12.0]
    for node in ("Hips", "Spine1"):
        out = orc.align_coeffs_to_previous_frame(coeffs, prev, joints, animated, node)
        assert out is not coeffs and out.shape == coeffs.shape
        np.testing.assert_allclose(out[0][[0, 2]], prev[[0, 2]], atol=1e-10)
        np.testing.assert_allclose(out[:, 1], coeffs[:, 1], atol=0)
        np.testing.assert_allclose(orc.node_heading(out[0], joints, animated, node), orc.node_heading(prev, joints, animated, node), atol=1e-12)
        np.testing.assert_allclose(np.linalg.norm(out[5][:3] - out[20][:3]), np.linalg.norm(coeffs[5][:3] - coeffs[20][:3]), rtol=1e-12)
        np.testing.assert_allclose(out[:, 7:], coeffs[:, 7:], atol=0)                     # only the root channels change
        again = orc.align_coeffs_to_previous_frame(out, out[0], joints, animated, node)  # already attached: identity
        np.testing.assert_allclose(again, out, atol=1e-9)
    # by hand: previous motion ends at (10, 0, 20) heading +x; the candidate starts at (1, 5, 2) heading +z and walks
    # 3 along +z -> aligned it starts at (10, 5, 20) and walks 3 along +x
    ident = np.zeros(3 + 4 * len(animated))
    ident[3::4] = 1.0
    prev2 = ident.copy()
    prev2[:3] = [10.0, 0.0, 20.0]
    prev2[3:7] = [np.sqrt(0.5), 0.0, np.sqrt(0.5), 0.0]        # +90 deg about y: ref (0,0,1) -> (1,0,0)
    walk = np.tile(ident, (4, 1))
    walk[:, :3] = [[1.0, 5.0, 2.0], [1.0, 5.0, 3.0], [1.0, 5.0, 4.0], [1.0, 5.0, 5.0]]
    out = orc.align_coeffs_to_previous_frame(walk, prev2, joints, animated, "Hips")
    np.testing.assert_allclose(out[:, :3], [[10.0, 5.0, 20.0], [11.0, 5.0, 20.0], [12.0, 5.0, 20.0], [13.0, 5.0, 20.0]], atol=1e-12)
    np.testing.assert_allclose(out[:, 3:7], np.tile(prev2[3:7], (4, 1)), atol=1e-12)


def test_two_hand_and_orientation_oracle_known_answers():
    """TwoHandConstraint residuals (reference two_hand_constraint.py:66-74) and the orientation distance
    (global_transform_constraint.py:109-121) on poses whose answers are known by hand."""
    from morphablegraphs_amd import synthetic
    joints, animated = synthetic.make_skeleton()
    frame = np.zeros(3 + 4 * len(animated))
    frame[3::4] = 1.0
    frame[:3] = [1.0, 2.0, 3.0]
    left = orc.joint_global_position(frame, joints, animated, "LeftHand_EndSite")
    right = orc.joint_global_position(frame, joints, animated, "RightHand_EndSite")
    np.testing.assert_allclose(left, [76.0, 35.0, 3.5], atol=1e-12)
    np.testing.assert_allclose(right, [-74.0, 35.0, 3.5], atol=1e-12)
    res = orc.two_hand_residuals(frame, joints, animated, ["LeftHand_EndSite", "RightHand_EndSite"], [[76.0, 35.0, 0.5], [-74.0, 39.0, 3.5]])
    np.testing.assert_allclose(res, [np.hypot(2.0, 1.5), 3.0, 4.0], atol=1e-12)
    mid = {"type": "joint_midpoint", "joint": "LeftHand_EndSite", "joint2": "RightHand_EndSite", "target": [1.0, 35.0, 0.5]}
    assert abs(orc.constraint_error_on_frame(mid, frame, joints, animated) - 3.0) < 1e-12
    half = np.sqrt(0.5)
    # identity pose against a wanted quarter turn about y: z axis vs x axis = pi / 2; against itself 0; a half turn about x = pi
    assert abs(orc.joint_orientation_error(frame, joints, animated, "LeftHand", [half, 0.0, half, 0.0]) - np.pi / 2) < 1e-12
    assert abs(orc.joint_orientation_error(frame, joints, animated, "LeftHand", [2.0, 0.0, 0.0, 0.0])) < 1e-12
    assert abs(orc.joint_orientation_error(frame, joints, animated, "Hips", [0.0, 1.0, 0.0, 0.0]) - np.pi) < 1e-7
    # the joint's OWN rotation counts, and so does every parent's: spine a quarter turn about y turns the hand's z axis to x
    f2 = frame.copy()
    ch = 3 + 4 * animated.index("Spine")
    f2[ch:ch + 4] = [half, 0.0, half, 0.0]
    assert abs(orc.joint_orientation_error(f2, joints, animated, "LeftHand", [half, 0.0, half, 0.0])) < 1e-12
    assert abs(orc.joint_orientation_error(f2, joints, animated, "Hips", [half, 0.0, half, 0.0]) - np.pi / 2) < 1e-12


def test_point_cloud_fit_is_optimal_and_recovers_a_known_transform():
    """The 2-D point-cloud fit behind the pose constraint (Kovar et al.'s closed form; anim_utils'
    align_point_clouds_2D is absent): it recovers a known rotation about y + xz translation exactly, and on clouds
    that do not match no nearby (theta, ox, oz) has a smaller weighted squared distance."""
    rng = np.random.default_rng(5)
    a = rng.standard_normal((12, 3)) * [30.0, 50.0, 30.0]
    w = rng.uniform(0.2, 2.0, 12)
    th, tx, tz = 0.7, 12.0, -5.0
    b = orc.transform_point_cloud(a, -th, 0.0, 0.0)                     # b = a turned back by th ...
    b[:, 0] -= tx * np.cos(th) - tz * np.sin(th)                        # ... and moved so that the fit must undo both
    b[:, 2] -= tx * np.sin(th) + tz * np.cos(th)
    theta, ox, oz = orc.align_point_clouds_2d(a, b, w)
    fitted = orc.transform_point_cloud(b, theta, ox, oz)
    np.testing.assert_allclose(fitted, a, atol=1e-10)
    assert abs(theta - th) < 1e-12

    def sse(cloud_b, t, x, z):
        f = orc.transform_point_cloud(cloud_b, t, x, z)
        return float((w * ((a[:, 0] - f[:, 0]) ** 2 + (a[:, 2] - f[:, 2]) ** 2)).sum())
    b2 = b + rng.standard_normal(b.shape) * 4.0                         # no exact fit any more
    t0, x0, z0 = orc.align_point_clouds_2d(a, b2, w)
    best = sse(b2, t0, x0, z0)
    for dt, dx, dz in ((1e-3, 0, 0), (-1e-3, 0, 0), (0, 1e-2, 0), (0, -1e-2, 0), (0, 0, 1e-2), (0, 0, -1e-2), (0.3, 1.0, -2.0)):
        assert sse(b2, t0 + dt, x0 + dx, z0 + dz) > best
    # the pose error of a cloud that is the wanted one turned and shifted is zero; a velocity of the first joint is added
    from morphablegraphs_amd import synthetic
    joints, animated = synthetic.make_skeleton()
    frame = np.zeros(3 + 4 * len(animated))
    frame[3::4] = 1.0
    frame[:3] = [1.0, 2.0, 3.0]
    names = ["Hips", "LeftHand", "RightHand", "Head", "LeftFoot"]
    cloud = np.array([orc.joint_global_position(frame, joints, animated, j) for j in names])
    wanted = orc.transform_point_cloud(cloud, 1.1, 40.0, -7.0)
    c = {"type": "pose", "joints": names, "points": wanted, "weights": [1.0, 2.0, 2.0, 0.5, 1.0], "velocity": None}
    assert orc.pose_constraint_error(c, frame, frame, joints, animated) < 1e-10
    frame2 = frame.copy()
    frame2[:3] += [0.5, 0.0, 2.0]
    c["velocity"] = [0.5, 0.0, 0.0]
    assert abs(orc.pose_constraint_error(c, frame, frame2, joints, animated) - 2.0) < 1e-10


def test_time_model_fixture():
    """The model with the legacy time part (reference motion_primitive.py:164-181): the mixture spans spatial + time
    latents, back_project(s, False) reads s[:n_s], and the canonical time function of s[n_s:]
    (_back_transform_gamma_to_canonical_time_function, :289-302) -- against vectors made by the reference itself."""
    from conftest import golden_model
    data, g = golden_model("time_model")
    prim = orc.OraclePrimitive(data)
    n_s, n_t = int(g["n_spatial_components"]), int(g["n_time_components"])
    assert (prim.n_components, prim.n_time_components) == (n_s, n_t) and g["S"].shape[1] == n_s + n_t
    np.testing.assert_allclose(prim.mean_temporal(), g["mean_temporal"], rtol=0, atol=1e-13)
    for s, ctf, fr in zip(g["S"], g["canonical_time_functions"], g["frames"]):
        np.testing.assert_allclose(prim.back_transform_gamma_to_canonical_time_function(s[n_s:]), ctf, rtol=1e-13, atol=1e-12)
        np.testing.assert_allclose(prim.back_project_frames(s), fr, rtol=0, atol=2e-12 * max(1.0, np.abs(fr).max()))
    np.testing.assert_allclose(prim.score_samples(g["S"]), g["logp_S"], rtol=1e-9, atol=1e-7)
    np.random.seed(int(g["seed"]))
    np.testing.assert_allclose(prim.sample_low_dimensional_vector(len(g["S"])), g["S"], rtol=1e-12, atol=1e-12)
    np.testing.assert_array_equal(g["low_dimensional_parameters"], g["S"][0])      # the spline keeps the full vector
    # the time-warped route (motion_primitive.py:206-234 with use_time_parameters=True, :268-319): vectors made by the reference's
    # own lines with np.linspace truncating its float sample count as NumPy <= 1.17 did (oracle/gen_golden.py)
    for tag, speed in (("speed10", 1.0), ("speed16", 1.6)):
        for b, s in enumerate(g["S"]):
            tf, fr = prim.back_project_warped_frames(s, speed)
            n = int(g["warped_lengths_" + tag][b])
            assert len(tf) == n
            np.testing.assert_allclose(tf, g["sample_time_functions_" + tag][b, :n], rtol=0, atol=1e-12)
            np.testing.assert_allclose(fr, g["warped_frames_" + tag][b, :n], rtol=0, atol=2e-12 * max(1.0, np.nanmax(np.abs(fr))))


def test_trajectory_spline_against_the_reference():
    """The Catmull-Rom spline under a trajectory constraint: points at parameters and the arc length of the
    granularity-1000 table, against vectors made by the reference's own ParameterizedSpline; and, on a smooth path that follows the
    spline loosely (one basin ahead of the bound), the reference's search and the monotone walk against each other."""
    from conftest import load_golden
    g = load_golden("trajectory_spline")
    for ci in range(int(g["n_cases"])):
        cps = g["control_points_%d" % ci]
        for u, pt in zip(g["parameters_%d" % ci], g["points_%d" % ci]):
            np.testing.assert_allclose(orc.catmull_rom_point(cps, u), pt, rtol=0, atol=1e-10)
        np.testing.assert_allclose(orc.catmull_rom_full_arc_length(cps), float(g["full_arc_length_%d" % ci]), rtol=1e-12)
        # the arc-length parameterisation the local / discrete / set trajectory constraints look their targets up by: the oracle's
        # table walk against the reference's query_point_by_absolute_arc_length
        table, full = orc.arc_length_table(cps)
        np.testing.assert_allclose(full, float(g["full_arc_length_%d" % ci]), rtol=1e-12)
        arcs, want = g["arc_lengths_%d" % ci], g["points_by_arc_%d" % ci]
        got = np.array([orc.catmull_rom_point_by_arc_length(cps, table, full, float(a)) for a in arcs])
        np.testing.assert_allclose(got, want, rtol=0, atol=1e-9)
    cps = g["control_points_1"]
    rng = np.random.default_rng(4)
    # a path that follows the spline loosely, as a root trajectory under a path-following constraint does
    us = np.linspace(0.02, 0.9, 40)
    path = np.array([orc.catmull_rom_point(cps, u) for u in us]) + rng.normal(0, 3.0, (40, 3)) * [1, 0, 1]
    min_u_a = min_u_b = 0.0
    for p in path:
        pa, ua = orc.closest_point_from(cps, p, min_u_a)
        pb, ub = orc.closest_point_walk(cps, p, min_u_b)
        da, db = np.linalg.norm(p - pa), np.linalg.norm(p - pb)
        assert abs(da - db) < 2e-3 * max(1.0, da), (ua, ub, da, db)
        min_u_a, min_u_b = ua, ub


def test_closest_point_search_against_the_references_own_run():
    """tests/golden/trajectory_closest_point.npz: ParameterizedSpline.find_closest_point_fast (parameterized_spline.py:303-322)
    run by the reference itself, chained frame to frame as trajectory_constraint.py:93-116 chains it, on 26 tracks.
    Held to it, two-sided, over whole chains of 156 frames:
      * the reference's call restated through scipy (closest_point_from): |du| <= 2e-7, |dd| <= 3e-7 -- what two evaluations of the
        same spline point that differ in the last bit do to a forward-difference gradient with h = 1e-8;
      * L-BFGS-B 3.0 restated for one bounded variable (lbfgsb_1d, what the device runs): the same bar -- with formk's bookkeeping
        transcribed, and with the rule the device uses in its place: identical parameters, bit for bit, on every track;
      * the monotone walk of rounds 2-4: equal where the distance has one basin ahead of the bound, far off elsewhere (why it is
        no longer the default)."""
    from conftest import load_golden
    g = load_golden("trajectory_closest_point")
    worst, far_walks, n_tracks = {"scipy": [0.0, 0.0], "lbfgsb": [0.0, 0.0]}, 0, 0
    for ci in range(int(g["n_cases"])):
        cps = g["control_points_%d" % ci]
        np.testing.assert_allclose(orc.catmull_rom_full_arc_length(cps), float(g["full_arc_length_%d" % ci]), rtol=1e-12)
        for ti, name in enumerate(g["track_names_%d" % ci]):
            track, u_ref, d_ref = g["track_%d_%d" % (ci, ti)], g["u_%d_%d" % (ci, ti)], g["distance_%d_%d" % (ci, ti)]
            stride = 1 if ci in (0, 3) else 4          # (every track whole where it is cheap; the chain needs no subsampling care: it restarts from the vectors)
            n_tracks += 1
            m_a = m_b = m_c = m_w = float(g["min_u0_%d_%d" % (ci, ti)])
            walk_far = False
            for f, p in enumerate(track):
                if stride > 1 and f >= 40:
                    break
                pa, m_a = orc.closest_point_from(cps, p, m_a)
                pb, m_b = orc.closest_point_lbfgsb(cps, p, m_b)
                pc, m_c = orc.closest_point_lbfgsb(cps, p, m_c, formk_rule=True)
                pw, m_w = orc.closest_point_walk(cps, p, m_w)
                assert m_c == m_b, (ci, name, f)
                for key, u, pt in (("scipy", m_a, pa), ("lbfgsb", m_b, pb)):
                    du, dd = abs(u - u_ref[f]), abs(np.linalg.norm(p - pt) - d_ref[f]) / max(1.0, d_ref[f])
                    worst[key][0], worst[key][1] = max(worst[key][0], du), max(worst[key][1], dd)
                    assert du <= 2e-7 and dd <= 3e-7, (key, ci, name, f, du, dd)
                walk_far = walk_far or abs(m_w - u_ref[f]) > 0.1
            far_walks += int(walk_far)
            # the target points the reference returned are the spline's points at its parameters
            np.testing.assert_allclose([orc.catmull_rom_point(cps, u) for u in u_ref[::13]], g["target_%d_%d" % (ci, ti)][::13], rtol=0, atol=1e-9)
    assert n_tracks == 26 and far_walks >= 5


def test_one_variable_lbfgsb_is_scipys():
    """lbfgsb_1d against scipy.optimize.minimize(method="L-BFGS-B") itself on seeded random problems (splines of 2..9 control points,
    points near and far, bounds anywhere): the same parameter to 1e-6 and the same iteration count, except where the forward
    difference's noise decides the last iteration (fewer than 1 search in 100; those agree to 2e-5)."""
    import math
    from scipy.optimize import minimize
    rng = np.random.default_rng(1)
    n, off, restarts = 0, 0, 0
    for trial in range(60):
        k = int(rng.integers(2, 10))
        cps = np.cumsum(np.column_stack([rng.uniform(5, 60, k), rng.uniform(-5, 5, k), rng.uniform(-40, 40, k)]), axis=0)
        for q in range(10):
            mu = float(rng.uniform(0, 1)) if rng.random() < 0.7 else float(rng.choice([0.0, 1.0 - 1e-9, 0.999, 1.0]))
            p = orc.catmull_rom_point(cps, rng.uniform(0, 1)) + rng.normal(0, rng.choice([0.1, 5, 50, 500]), 3)

            def dist(x):
                v = orc.catmull_rom_point(cps, float(np.ravel(x)[0])) - p
                return math.sqrt(float(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]))
            res = minimize(dist, np.array([mu]), method="L-BFGS-B", bounds=[(mu, 1.0)])
            info = {}
            u = orc.lbfgsb_1d(dist, mu, mu, 1.0, info=info)
            n += 1
            restarts += info["formk_restarts"]
            du = abs(u - float(res.x[0]))
            assert du <= 2e-5, (trial, q, u, res.x)
            if du > 1e-6 or info["nit"] != res.get("nit", 0):
                off += 1
    assert n == 600 and off <= 6, off
    assert restarts > 20          # (formk's failures are part of the algorithm: one search in eight meets one)


def test_device_search_statements_on_the_host_against_the_references_vectors():
    """The DEVICE's closest-point search -- the statements of morphablegraphs_amd/csrc/mg_traj_device.h (mg_traj_closest_lbfgsb),
    compiled for the host by tests/native/Makefile -- on the golden tracks: every chain within the search's stated tolerance of what
    the reference's own function returned (|du| <= 2e-6, |dd| <= 1e-6 max(1, d); measured 1.7e-7 / 3e-9), and within the forward
    difference's noise of the oracle's restatement it mirrors.  (No GPU: the header's arithmetic is plain C++ double arithmetic
    without contraction on both sides; tests/test_gpu_closest_point.py runs the same vectors on the device.)"""
    import ctypes as C
    import os
    import shutil
    import subprocess
    from conftest import load_golden
    native = os.path.join(os.path.dirname(os.path.abspath(__file__)), "native")
    if shutil.which("hipcc") is None and not os.path.exists("/opt/rocm/bin/hipcc"):
        pytest.skip("no hipcc to build the host check")
    subprocess.check_call(["make", "-s", "-C", native])
    lib = C.CDLL(os.path.join(native, "liblbfgsb_host_check.so"))
    lib.mg_test_lbfgsb_chain.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_double, C.c_void_p, C.c_void_p]
    g = load_golden("trajectory_closest_point")
    worst_u = worst_d = 0.0
    for ci in range(int(g["n_cases"])):
        cps = np.ascontiguousarray(g["control_points_%d" % ci])
        for ti, name in enumerate(g["track_names_%d" % ci]):
            track = np.ascontiguousarray(g["track_%d_%d" % (ci, ti)])
            u_ref, d_ref, mu = g["u_%d_%d" % (ci, ti)], g["distance_%d_%d" % (ci, ti)], float(g["min_u0_%d_%d" % (ci, ti)])
            u, d = np.empty(len(track)), np.empty(len(track))
            assert lib.mg_test_lbfgsb_chain(cps.ctypes.data, len(cps), track.ctypes.data, len(track), mu, u.ctypes.data, d.ctypes.data) == 0
            du, dd = np.abs(u - u_ref).max(), (np.abs(d - d_ref) / np.maximum(1.0, d_ref)).max()
            assert du <= 2e-6 and dd <= 1e-6, (ci, name, du, dd)
            worst_u, worst_d = max(worst_u, du), max(worst_d, dd)
            if ci == 1:      # ... and the oracle's restatement with the device's formk rule, frame by frame from the device's own chain
                for f in range(0, len(track), 7):
                    _, uo = orc.closest_point_lbfgsb(cps, track[f], mu if f == 0 else u[f - 1], formk_rule=True)
                    assert abs(uo - u[f]) <= 2e-6, (name, f, uo, u[f])
    assert worst_u <= 5e-7


def test_per_frame_classes_against_the_references_own_classes():
    """tests/golden/per_frame_classes.npz: LocalTrajectoryConstraint (local_trajectory_constraint.py:45-78) and TrajectorySetConstraint
    (trajectory_set_constraint.py:41-104) run by the reference itself on given joint tracks (a duck-typed skeleton that returns
    them: input data); the oracle's restatements downstream of forward kinematics against them."""
    from conftest import load_golden
    g = load_golden("per_frame_classes")
    for ci in range(int(g["n_cases"])):
        cps, cps2, hips, hand = g["control_points_%d" % ci], g["control_points2_%d" % ci], g["hips_%d" % ci], g["hand_%d" % ci]
        for si in range(2):
            nf = int(g["local_n_frames_%d_%d" % (ci, si)])
            c = {"type": "frame_local_trajectory", "joint": "Hips", "control_points": cps.tolist(), "granularity": 1000,
                 "start_t": float(g["local_start_t_%d_%d" % (ci, si)]), "n_frames": nf}
            r, e = orc.per_frame_track_residuals(c, {"Hips": hips[:nf]})
            want = g["local_residuals_%d_%d" % (ci, si)]
            np.testing.assert_allclose(r, want, rtol=1e-9, atol=1e-9 * max(1.0, want.max()))
            assert abs(e - float(g["local_error_%d_%d" % (ci, si)])) <= 1e-9 * max(1.0, abs(e))
        for si in range(3):
            rg = g["set_ranges_%d_%d" % (ci, si)]
            trs = [{"control_points": cp.tolist(), "granularity": 1000, "range_start": None if np.isnan(r[0]) else float(r[0]),
                    "range_end": None if np.isnan(r[1]) else float(r[1])} for cp, r in zip((cps, cps2), rg)]
            c = {"type": "frame_trajectory_set", "joints": ["Hips", "LeftHand"], "trajectories": trs,
                 "arc_lengths": g["set_arc_lengths_%d_%d" % (ci, si)].tolist(), "n_frames": len(hips)}
            r, e = orc.per_frame_track_residuals(c, {"Hips": hips, "LeftHand": hand})
            want = g["set_residuals_%d_%d" % (ci, si)]
            np.testing.assert_allclose(r, want, rtol=1e-10, atol=1e-10 * max(1.0, want.max()))
            assert abs(e - float(g["set_error_%d_%d" % (ci, si)])) <= 1e-10 * max(1.0, abs(e))
            assert (si == 0) == (not want.any())


def test_keyframe_classes_against_the_references_own_classes():
    """tests/golden/keyframe_classes.npz: TwoHandConstraintSet (two_hand_constraint.py:33-93) and FeetConstraint
    (feet_constraint.py:30-55) run by the reference itself on its own MotionSpline objects of the walk model (forward kinematics:
    the oracle's, SELF-DEFINED -- the fixture stores the hand positions it produced); the oracle's restatements on the oracle's
    frames against them."""
    from conftest import load_golden
    from morphablegraphs_amd import synthetic
    g = load_golden("keyframe_classes")
    data = synthetic.make_walk_primitive(seed=0)
    joints, animated = synthetic.make_skeleton()
    op = orc.OraclePrimitive(data)
    S = g["S"]
    coeffs = [op.back_project_spatial_coeffs(s) for s in S]
    for ci in range(int(g["n_two_hand"])):
        key, names, positions = float(g["two_hand_keyframe_%d" % ci]), [str(n) for n in g["two_hand_joints_%d" % ci]], g["two_hand_positions_%d" % ci]
        want, hands = g["two_hand_residuals_%d" % ci], g["two_hand_hand_positions_%d" % ci]
        for b in range(len(S)):
            frame = orc.spline_frames(op.knots, coeffs[b], [key])[0]
            for k in range(2):
                np.testing.assert_allclose(orc.joint_global_position(frame, joints, animated, names[k]), hands[b, k], rtol=0, atol=1e-9)
            r = orc.two_hand_residuals(frame, joints, animated, names, positions)
            np.testing.assert_allclose(r, want[b], rtol=1e-10, atol=1e-9)
            assert abs(sum(r) - g["two_hand_error_%d" % ci][b]) <= 1e-9 * max(1.0, sum(r))
    for ci in range(int(g["n_feet"])):
        key, w = float(g["feet_keyframe_%d" % ci]), float(g["feet_weight_%d" % ci])
        for b in range(len(S)):
            frame = orc.spline_frames(op.knots, coeffs[b], [key])[0]
            r = orc.feet_residuals(frame, joints, animated, g["feet_left_%d" % ci], g["feet_right_%d" % ci], w)
            np.testing.assert_allclose(r, g["feet_residuals_%d" % ci][b], rtol=1e-10, atol=1e-9)
            assert abs(sum(r) - g["feet_residuals_spline_%d" % ci][b, 0]) <= 1e-9 * max(1.0, sum(r))
            assert abs(sum(r) - g["feet_error_%d" % ci][b]) <= 1e-9 * max(1.0, sum(r))


def test_time_constraints_against_the_references_own_class():
    """tests/golden/time_constraints.npz: TimeConstraints (time_constraints.py:25-110) run by the reference itself over a
    three-step walk on the time_model primitive (oracle/gen_golden.py run_time_constraints_case); the oracle's restatement over
    the oracle's canonical time functions and mixture against it."""
    from conftest import golden_model, load_golden
    data, gm = golden_model("time_model")
    g = load_golden("time_constraints")
    assert str(g["digest"]) == str(gm["digest"])
    np.testing.assert_array_equal(g["base"], gm["S"][:3])
    op = orc.OraclePrimitive(data)
    op.init_time_model(data)
    n_s, n_t = int(gm["n_spatial_components"]), int(gm["n_time_components"])
    base, frame_time = g["base"], float(g["frame_time"])
    varied = 0
    for ci in range(int(g["n_cases"])):
        start, end = int(g["start_step_%d" % ci]), int(g["end_step_%d" % ci])
        before = [op.back_transform_gamma_to_canonical_time_function(base[i][n_s:]) for i in range(start)]
        start_keyframe = orc.time_constraints_start_keyframe(before)
        assert abs(start_keyframe - float(g["start_keyframe_%d" % ci])) <= 1e-9
        np.testing.assert_array_equal(np.concatenate([base[i][n_s:] for i in range(start, end)]), g["initial_guess_%d" % ci])
        clist = [tuple(row) for row in g["constraint_list_%d" % ci]]
        for b, s in enumerate(g["S_%d" % ci]):
            tfs = [op.back_transform_gamma_to_canonical_time_function(s[k * n_t:(k + 1) * n_t]) for k in range(end - start)]
            e = orc.time_constraints_error(tfs, clist, start_keyframe, frame_time)
            want = float(g["error_%d" % ci][b])
            assert abs(e - want) <= 1e-9 * max(1.0, abs(want)), (ci, b, e, want)
            ll = np.mean([op.score_samples(np.concatenate([base[start + k][:n_s], s[k * n_t:(k + 1) * n_t]])[None, :])[0] for k in range(end - start)])
            assert abs(ll - float(g["loglikelihood_%d" % ci][b])) <= 1e-9 * max(1.0, abs(ll))
        varied += len(np.unique(np.round(g["error_%d" % ci], 9))) > 1
    assert varied >= 2                                                   # the candidates' time functions do move the error


def test_walk_32_fixture_configs0():
    """BASELINE configs[0] at its stated size: one 'walk' primitive, 32 latent samples (the reference drew and back-projected them:
    oracle/gen_golden.py run_walk_32_case) -- the oracle's NumPy path and the C restatement against the reference's frames."""
    from conftest import load_golden
    from morphablegraphs_amd import synthetic
    from oracle import c_oracle
    g = load_golden("walk_32")
    data = synthetic.make_walk_primitive(seed=0)
    prim = orc.OraclePrimitive(data)
    S, rows, want = g["S"], g["frame_rows"], g["frames_at_rows"]
    assert S.shape == (32, 40) and want.shape == (32, len(rows), 79)
    np.random.seed(int(g["seed"]))
    np.testing.assert_allclose(prim.sample_low_dimensional_vector(32), S, rtol=1e-12, atol=1e-12)
    scale = max(1.0, np.abs(want).max())
    got = prim.back_project_frames_batch(S)[:, rows]
    np.testing.assert_allclose(got, want, rtol=0, atol=2e-12 * scale)
    cp = c_oracle.COraclePrimitive(data)
    np.testing.assert_allclose(cp.frames_f64(S)[:, rows], want, rtol=0, atol=2e-12 * scale)
    np.testing.assert_allclose(cp.log_prob_f64(S), g["logp"], rtol=1e-9, atol=1e-7)
    m32 = cp.frames_f32model(S)[:, rows].astype(np.float64)
    assert np.all(np.abs(m32 - want) <= 1e-5 + 2.0 ** -24 * np.abs(want))


@pytest.mark.skipif(not os.path.isdir("/root/reference/morphablegraphs"), reason="the reference is only in the build container")
def test_fixtures_reproduce_from_the_reference():
    """Every tests/golden/*.npz regenerated from /root/reference by oracle/gen_golden.py (oracle/check_golden.py): each array
    equal to the committed one."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "oracle", "check_golden.py")], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "13 fixture(s) regenerated, 0 array(s) differ" in r.stdout, r.stdout
