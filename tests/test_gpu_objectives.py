"""The graph-walk (global) objectives, the time objective, the step-goal objectives and the scipy.minimize route, batched
(reference optimization/objective_functions.py:59-140,270-380; numerical_minimizer.py:41-76; constraints/time_constraints.py).
Every value is compared with a per-candidate restatement of the reference's loop built from the oracle."""
import numpy as np
import pytest

from morphablegraphs_amd import _capi, synthetic
from morphablegraphs_amd import objective_functions as of
from morphablegraphs_amd.motion_state_graph import HipMotionStateGraphNode
from oracle import mg_oracle as orc

pytestmark = pytest.mark.gpu


class _Skeleton(object):
    def __init__(self, node, frame_time=1.0 / 30.0):
        self.aligning_root_node, self.aligning_root_dir, self.root, self.frame_time = node, (0.0, 0.0, 1.0), "Hips", frame_time


class _Constraints(object):
    def __init__(self, cons, is_local, hip_sk, ref_sk):
        self.constraints, self.is_local, self.hip_skeleton, self.skeleton = cons, is_local, hip_sk, ref_sk
        self.start_pose, self.min_error, self.evaluations = None, None, 0


class _Step(object):
    def __init__(self, key, parameters, n_spatial, n_time, cons):
        self.node_key, self.parameters, self.n_spatial_components, self.n_time_components = key, np.asarray(parameters, dtype=np.float64), n_spatial, n_time
        self.motion_primitive_constraints = cons


class _Graph(object):
    def __init__(self, nodes, ref_sk, hip_sk):
        self.nodes, self.skeleton, self.hip_skeleton = nodes, ref_sk, hip_sk


def _walk(aligning_node, local_step=None):
    joints, animated = synthetic.make_skeleton()
    hip_sk = _capi.Skeleton(joints, animated)
    ref_sk = _Skeleton(aligning_node)
    shapes = [(40, 156, 8), (24, 60, 3), (33, 97, 5)]
    nodes, ops, steps, datas = {}, [], [], []
    rng = np.random.default_rng(5)
    for i, (L, F, K) in enumerate(shapes):
        data = synthetic.make_primitive(seed=40 + i, n_components=L, n_frames=F, n_gmm=K, name="m%d" % i)
        node = HipMotionStateGraphNode()
        node.init_from_dict("walk", {"name": "m%d" % i, "mm": data})
        nodes[node.node_key] = node
        ops.append(orc.OraclePrimitive(data))
        datas.append(data)
        tl = float(F - 1)
        cons = [{"type": "position", "t": tl, "weight": 1.0, "target": [30.0 * (i + 1), None, -20.0 * i]},
                {"type": "direction", "t": tl / 2.0, "weight": 0.5, "target": [0.3, 1.0]},
                {"type": "joint_position", "joint": "LeftHand", "t": tl, "weight": 2.0, "target": [25.0 * (i + 1), 95.0, -15.0 * i]}]
        steps.append(_Step(node.node_key, rng.standard_normal(L), L, 0, _Constraints(cons, local_step == i, hip_sk, ref_sk)))
    return _Graph(nodes, ref_sk, hip_sk), steps, ops, (joints, animated)


@pytest.mark.parametrize("aligning_node,local_step,with_prev", [("Hips", None, True), ("Spine", 1, True), ("Hips", None, False)])
def test_global_graph_walk_objectives_chain_every_candidate_to_its_own_previous_step(aligning_node, local_step, with_prev):
    graph, steps, ops, (joints, animated) = _walk(aligning_node, local_step)
    rng = np.random.default_rng(17)
    Ls = [st.n_spatial_components for st in steps]
    n = 7
    S = 0.7 * rng.standard_normal((n, sum(Ls)))
    prev_frames = None
    if with_prev:
        prev_frames = ops[0].back_project_frames(rng.standard_normal(Ls[0]))[-3:].copy()
        prev_frames[:, 0] += 120.0
        prev_frames[:, 2] -= 40.0
    cons = [st.motion_primitive_constraints.constraints for st in steps]
    local = () if local_step is None else (local_step,)

    def chain(row, exit_from):
        alphas, off = [], 0
        for L in Ls:
            alphas.append(row[off:off + L])
            off += L
        return orc.graph_walk_residual_blocks(ops, alphas, cons, None if prev_frames is None else prev_frames[-1], joints, animated, aligning_node,
                                              exit_from=exit_from, local_steps=local)
    # obj_global_error_sum: a float for one vector, (n,) for a batch
    data = (graph, steps, 1.0, 0.1, prev_frames)
    total = of.obj_global_error_sum(S, data)
    ref = np.array([sum(b.sum() for b in chain(row, "frames")) for row in S])
    np.testing.assert_allclose(total, ref, rtol=1e-9, atol=1e-9)
    assert abs(of.obj_global_error_sum(S[2], data) - ref[2]) <= 1e-9 * max(1.0, abs(ref[2]))
    # residual vector: the steps' residuals, each padded to its number of variables, divided by init_error_sum
    init = 3.5
    rv = of.obj_global_residual_vector(S, data + (init,))
    assert rv.shape == (n, sum(Ls))
    for b, row in enumerate(S):
        cols = [np.concatenate([blk, np.zeros(L - len(blk))]) for blk, L in zip(chain(row, "frames"), Ls)]
        np.testing.assert_allclose(rv[b], np.concatenate(cols) / init, rtol=1e-9, atol=1e-9)
    # ... and naturalness: residual * error_scale - log p * quality_scale; the chain runs through the aligned CONTROL POINTS
    error_scale, quality_scale = 0.8, 0.05
    rn = of.obj_global_residual_vector_and_naturalness(S, (graph, steps, error_scale, quality_scale, prev_frames, init))
    for b, row in enumerate(S):
        cols, off = [], 0
        for blk, L, op in zip(chain(row, "coeffs"), Ls, ops):
            nll = -op.score_samples(row[off:off + L][None, :])[0] * quality_scale
            off += L
            cols.append(np.concatenate([blk * error_scale + nll, np.zeros(L - len(blk))]))
        np.testing.assert_allclose(rn[b], np.concatenate(cols) / init, rtol=1e-8, atol=1e-8)
    # the chain is real: changing only the FIRST step's latents changes the LAST step's residuals (when anything is aligned)
    S2 = S.copy()
    S2[:, :Ls[0]] += 0.3
    rv2 = of.obj_global_residual_vector(S2, data + (init,))
    assert not np.allclose(rv2[:, Ls[0] + Ls[1]:], rv[:, Ls[0] + Ls[1]:])
    for node in graph.nodes.values():
        node.motion_primitive._prim.close()


def test_step_goal_objectives_and_the_minimizer_route():
    """step_goal_error / _jac (+ naturalness) against the last control point of back_project_spatial_coeffs and finite
    differences; HipNumericalMinimizer: scipy.minimize with ONE launch per gradient (the objective over len(s) + 1 points)
    lowers the objective like scipy's own sequential differences on the oracle's objective."""
    from scipy.optimize import minimize
    from morphablegraphs_amd.motion_primitive_generator import HipNumericalMinimizer, HipOptimizerBuilder
    data = synthetic.make_walk_primitive(seed=0)
    node = HipMotionStateGraphNode()
    node.init_from_dict("walk", {"name": "leftStance", "mm": data})
    op = orc.OraclePrimitive(data)

    class Goal(object):
        position = np.array([35.0, 0.0, -12.0])
    cons = _Constraints([Goal()], True, None, None)
    rng = np.random.default_rng(2)
    S = rng.standard_normal((5, 40))
    err = of.step_goal_error(S, (node, cons, None))
    for b, s in enumerate(S):
        pos = op.back_project_spatial_coeffs(s)[-1, :3].copy()
        pos[1] = 0.0
        assert abs(err[b] - np.dot(Goal.position - pos, Goal.position - pos)) <= 1e-9 * max(1.0, err[b])
    jac = of.step_goal_jac(S, (node, cons, None))
    eps = 1e-6
    for i in (0, 7, 39):
        Sp = S.copy()
        Sp[:, i] += eps
        np.testing.assert_allclose(jac[:, i], (of.step_goal_error(Sp, (node, cons, None)) - err) / eps, rtol=1e-4, atol=1e-4)
    jn = of.step_goal_and_naturalness_jac(S, (node, cons, None))
    np.testing.assert_allclose(jn, jac - np.array([op.log_likelihood_jac(s[None, :])[0] for s in S]), rtol=1e-8, atol=1e-8)
    settings = {"method": "BFGS", "max_iterations": 30, "tolerance": 1e-8, "diff_eps": 1e-7, "verbose": False,
                "start_error_threshold": 0.0, "error_scale_factor": 1.0, "quality_scale_factor": 0.05}
    # path-following minimizer: analytic Jacobian, the goal is met
    m = HipOptimizerBuilder({"local_optimization_settings": settings}).build_path_following_minimizer()
    m.set_objective_function_parameters((node, cons, None))
    x = m.run(S[0])
    assert of.step_goal_error(x, (node, cons, None)) < 1e-6 * max(1.0, err[0])
    # spatial error + naturalness without a Jacobian: batched forward differences, one launch per gradient
    kcons = _Constraints([{"type": "position", "t": 155.0, "weight": 1.0, "target": [40.0, None, -30.0]},
                          {"type": "direction", "t": 155.0, "weight": 1.0, "target": [0.5, 1.0]}], True, None, None)
    mm = HipNumericalMinimizer(settings, of.obj_spatial_error_sum_and_naturalness)
    params = (node, kcons, None, 1.0, 0.05, 1.0)
    mm.set_objective_function_parameters(params)
    x0 = S[1]
    f0 = of.obj_spatial_error_sum_and_naturalness(x0, params)
    x1 = mm.run(x0)
    f1 = of.obj_spatial_error_sum_and_naturalness(x1, params)

    def oracle_obj(s):
        res = op.keyframe_residuals(s[None, :], [{"type": "position", "t": 155.0, "weight": 1.0, "target": [40.0, None, -30.0]},
                                                 {"type": "direction", "t": 155.0, "weight": 1.0, "target": [0.5, 1.0]}])[0]
        return float(res.sum() - 0.05 * op.score_samples(s[None, :])[0])
    assert abs(f0 - oracle_obj(x0)) <= 1e-9 * max(1.0, abs(f0))
    ref = minimize(oracle_obj, x0, method="BFGS", tol=1e-8, options={"maxiter": 30, "eps": 1e-7})
    assert f1 < f0 and abs(f1 - ref.fun) <= 2e-2 * max(1.0, abs(ref.fun))      # same method, same step: the same basin, to the line searches' noise
    assert mm.n_launches < 6 * 30 + 10                                            # not (L + 1) objective calls per gradient
    node.motion_primitive._prim.close()


def test_time_objective_for_batches():
    """obj_time_error_sum / HipTimeConstraints against the reference's per-candidate loop (time_constraints.py:40-102) over the
    oracle's canonical time functions and mixture."""
    from conftest import golden_model
    data, g = golden_model("time_model")
    n_s, n_t = int(g["n_spatial_components"]), int(g["n_time_components"])
    node = HipMotionStateGraphNode()
    node.init_from_dict("walk", {"name": "tm", "mm": data})
    op = orc.OraclePrimitive(data)
    op.init_time_model(data)
    rng = np.random.default_rng(4)
    base = [np.asarray(g["S"][i], dtype=np.float64) for i in range(3)]
    steps = [_Step(node.node_key, b, n_s, n_t, None) for b in base]

    class Walk(object):
        pass
    walk = Walk()
    walk.steps = steps
    graph = _Graph({node.node_key: node}, _Skeleton("Hips", frame_time=0.02), None)
    F = int(data["n_canonical_frames"])
    constraint_list = [(0, F // 2, 1.1), (1, F - 1, 3.9), (2, 5, 4.2), (5, 1, 1.0), (1, F + 3, 2.0)]
    tc = of.HipTimeConstraints(graph, walk, 1, 3, constraint_list)          # optimise steps 1 and 2; step 0 fixes the start frame
    start_ref = op.back_transform_gamma_to_canonical_time_function(base[0][n_s:])[-1]
    assert abs(tc.start_keyframe - start_ref) <= 1e-9
    S = 0.5 * rng.standard_normal((6, 2 * n_t))
    got = of.obj_time_error_sum(S, (graph, walk, tc, 2.0, 0.3))
    for b, s in enumerate(S):
        tfs = [op.back_transform_gamma_to_canonical_time_function(s[k * n_t:(k + 1) * n_t]) for k in range(2)]
        err = 0.0
        for step_index, key, desired in constraint_list:
            n_frames, e = start_ref, 10000.0
            for k, tf in enumerate(tfs):
                if k < step_index:
                    n_frames += tf[-1]
                else:
                    e = 0.0 if key >= len(tf) else (desired - (n_frames + int(tf[key]) + 1) * 0.02) ** 2
                    break
            err += e
        ll = np.mean([op.score_samples(np.concatenate([base[1 + k][:n_s], s[k * n_t:(k + 1) * n_t]])[None, :])[0] for k in range(2)])
        assert abs(got[b] - (2.0 * err - 0.3 * ll)) <= 1e-8 * max(1.0, abs(got[b])), (b, got[b], 2.0 * err - 0.3 * ll)
    assert len(tc.get_initial_guess(walk)) == 2 * n_t
    node.motion_primitive._prim.close()


def test_time_constraints_against_the_references_own_class():
    """tests/golden/time_constraints.npz: TimeConstraints run by the reference itself (time_constraints.py:25-110) over a
    three-step walk on the time_model primitive -- start frame, initial guess, the error of every candidate, the average
    log-likelihood (its [0] on the mixture's score read as the old per-sample score, see oracle/gen_golden.py)."""
    from conftest import golden_model, load_golden
    data, gm = golden_model("time_model")
    g = load_golden("time_constraints")
    n_s, n_t = int(gm["n_spatial_components"]), int(gm["n_time_components"])
    node = HipMotionStateGraphNode()
    node.init_from_dict("walk", {"name": "tm", "mm": data})

    class Walk(object):
        pass
    walk = Walk()
    walk.steps = [_Step(node.node_key, b, n_s, n_t, None) for b in g["base"]]
    graph = _Graph({node.node_key: node}, _Skeleton("Hips", frame_time=float(g["frame_time"])), None)
    for ci in range(int(g["n_cases"])):
        clist = [(int(r[0]), int(r[1]), float(r[2])) for r in g["constraint_list_%d" % ci]]
        tc = of.HipTimeConstraints(graph, walk, int(g["start_step_%d" % ci]), int(g["end_step_%d" % ci]), clist)
        assert abs(tc.start_keyframe - float(g["start_keyframe_%d" % ci])) <= 1e-9
        np.testing.assert_array_equal(tc.get_initial_guess(walk), g["initial_guess_%d" % ci])
        S = g["S_%d" % ci]
        want_e, want_l = g["error_%d" % ci], g["loglikelihood_%d" % ci]
        np.testing.assert_allclose(tc.evaluate_graph_walk(S, graph, walk), want_e, rtol=1e-9, atol=1e-9)
        np.testing.assert_allclose(tc.get_average_loglikelihood(S, graph, walk), want_l, rtol=1e-9, atol=1e-8)
        np.testing.assert_allclose(of.obj_time_error_sum(S, (graph, walk, tc, 2.0, 0.3)), 2.0 * want_e - 0.3 * want_l, rtol=1e-9, atol=1e-8)
        assert abs(tc.evaluate_graph_walk(S[0], graph, walk) - want_e[0]) <= 1e-9 * max(1.0, want_e[0])      # one candidate: a float
    node.motion_primitive._prim.close()


@pytest.mark.parametrize("n", [17, 4099, 40000, 131072 + 5])   # (the last: BASELINE configs[4]'s iteration -- waves with two and three tiles, a ragged last one)
def test_objective_in_one_launch_is_bit_identical_to_the_two_calls(n):
    """mg_objective_error_and_naturalness: the LDS-resident mixture kernel scores the keyframe constraints on
    the latent tile it holds and writes error_scale * error + quality_scale * (-log p).  The three outputs must be the bits of
    mg_score_constraints, mg_gmm_log_prob (float64) and of NumPy's array arithmetic on them -- root and forward-kinematics
    constraints, local and aligned to a previous frame, float32 and float64 latents, ragged last tile."""
    from morphablegraphs_amd import _capi, synthetic
    from morphablegraphs_amd.candidate_scoring import alignment_from_prev_frames
    ctx = _capi.Context(0)
    data = synthetic.make_walk_primitive(seed=0)
    prim = _capi.Primitive(ctx, data)
    joints, animated = synthetic.make_skeleton(19)
    sk = _capi.Skeleton(joints, animated)
    cons_root = [{"type": "position", "t": 155.0, "weight": 1.0, "target": [40.0, None, -30.0]},
                 {"type": "direction", "t": 155.0, "weight": 1.0, "target": [0.5, 1.0]}]
    cons_fk = cons_root + [{"type": "joint_position", "joint": "RightHand", "t": 77.5, "weight": 2.0, "target": [12.0, 90.0, 4.0]}]
    prev = np.random.default_rng(1).standard_normal((3, 79))
    prev[:, 3:7] = [1.0, 0.1, 0.0, 0.05]
    rng = np.random.default_rng(n)
    for clist, skeleton, alignment in ((cons_root, None, None), (cons_fk, sk, None), (cons_root, None, sk.alignment_to(prev[-1]))):
        cset = _capi.ConstraintSet(prim, clist, skeleton, alignment)
        if skeleton is not None:      # joint constraints: not on the one-launch kernel (it has no registers for the chains) -- it must say so
            with pytest.raises(_capi.MGError) as e:
                prim.objective(cset, np.zeros((4, 40)), 0.75, 1.25)
            assert e.value.status == -4
            cset.close()
            continue
        for dtype in (np.float32, np.float64):
            S = (1.5 * rng.standard_normal((n, 40))).astype(dtype)
            obj, err, lp = prim.objective(cset, S, 0.75, 1.25)
            err2 = prim.score_constraints(cset, S)
            lp2 = prim.gmm_log_prob(S.astype(np.float64) if dtype == np.float64 else S, dtype=np.float64)
            np.testing.assert_array_equal(err.view(np.uint64), err2.view(np.uint64))
            np.testing.assert_array_equal(lp.view(np.uint64), lp2.view(np.uint64))
            np.testing.assert_array_equal(obj.view(np.uint64), (0.75 * err2 + (-lp2) * 1.25).view(np.uint64))
        cset.close()
    prim.close()
    ctx.close()


def test_naturalness_objective_takes_the_one_launch_route_and_agrees_with_the_two_calls(monkeypatch):
    """obj_spatial_error_sum_and_naturalness uses mg_objective_error_and_naturalness for root constraint sets; with that route
    switched off it makes the two calls -- the same values (the sum over constraints is NumPy's there: 1e-12)."""
    from morphablegraphs_amd import objective_functions as of, synthetic
    from morphablegraphs_amd.motion_primitive import HipMotionPrimitive
    prim = HipMotionPrimitive(None)
    prim._initialize_from_json(synthetic.make_walk_primitive(seed=0))
    cons = [{"type": "position", "t": 155.0, "weight": 1.0, "target": [40.0, None, -30.0]},
            {"type": "direction", "t": 100.0, "weight": 0.5, "target": [0.5, 1.0]}]
    S = 1.3 * np.random.default_rng(2).standard_normal((3000, 40))
    data = (prim, cons, None, 0.8, 1.7, 1.0)      # (..., error_scale, quality_scale, init_error_sum): the objective reads data[-3], data[-2]
    calls = []
    real = of._objective_in_one_launch
    monkeypatch.setattr(of, "_objective_in_one_launch", lambda *a: calls.append(1) or real(*a))
    a = of.obj_spatial_error_sum_and_naturalness(S, data)
    assert calls and real(of._prim_of(prim), cons, S, None, 0.8, 1.7) is not None       # the one-launch route was taken
    monkeypatch.setattr(of, "_objective_in_one_launch", lambda *a: None)
    b = of.obj_spatial_error_sum_and_naturalness(S, data)
    np.testing.assert_allclose(a, b, rtol=1e-12, atol=1e-12)
    assert isinstance(of.obj_spatial_error_sum_and_naturalness(S[0], data), float)
