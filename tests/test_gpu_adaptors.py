"""GPU tests of the reference-shaped Python surface (HipMotionPrimitive / HipMotionSpline /
HipGaussianMixture / wrapper / batched candidate scoring) against the golden vectors made by the
reference's own code and against the oracle.  The call scripts read like the reference's callers."""
import numpy as np
import pytest

from morphablegraphs_amd import (HipMotionPrimitive, HipMotionPrimitiveModelWrapper, _capi, synthetic)
from morphablegraphs_amd.candidate_scoring import HipSampleFilter, evaluate_samples_using_constraints
from morphablegraphs_amd.motion_state_graph import HipMotionStateGraphNode, HipPrimitiveSet, arc_length_xz
from oracle import c_oracle

pytestmark = pytest.mark.gpu


def pose_tol(ref):
    return 1e-5 + 2.0 ** -24 * np.abs(ref)


def _primitive(data):
    p = HipMotionPrimitive(None)
    p._initialize_from_json(data)
    return p


def test_reference_call_script_on_golden(golden_case):
    """examples/run_construction.py load_model shape: load JSON -> sample(False) -> get_motion_vector()."""
    name, data, g = golden_case
    mp = _primitive(data)
    scale = max(1.0, np.abs(g["frames"]).max())
    # sampling consumes numpy's global stream exactly like sklearn's GaussianMixture.sample
    np.random.seed(int(g["seed"]))
    S = mp.sample_low_dimensional_vector(g["S"].shape[0])
    np.testing.assert_allclose(S, g["S"], rtol=1e-12, atol=1e-12)
    for b in range(len(S)):
        spline = mp.back_project(g["S"][b], use_time_parameters=False)
        np.testing.assert_array_equal(spline.time_function, g["time_function"])
        np.testing.assert_array_equal(np.asarray(spline.knots), g["knots"])
        np.testing.assert_allclose(spline.coeffs, g["coeffs"][b], rtol=0, atol=4e-12 * scale)
        frames = spline.get_motion_vector()
        assert frames.shape == g["frames"][b].shape and frames.dtype == np.float64
        np.testing.assert_allclose(frames, g["frames"][b], rtol=0, atol=4e-12 * scale)
        np.testing.assert_allclose(spline.evaluate(g["eval_times"]), g["evals"][b], rtol=0, atol=4e-12 * scale)
    sp0 = mp.back_project(g["S"][0], False)
    for i, t in enumerate(g["eval_times"]):
        e = sp0.evaluate(float(t))
        assert e.shape == (mp.s_pca["n_dim"],)
        np.testing.assert_allclose(e, g["evals_scalar"][i], rtol=0, atol=4e-12 * scale)
    assert mp.get_n_canonical_frames() == int(g["n_canonical_frames"])
    assert mp.get_n_spatial_components() == int(g["n_spatial_components"])
    assert mp.get_n_time_components() == 0 and not mp.has_time_parameters
    # batched hot path against the same golden frames
    fb = mp.back_project_frames_batch(g["S"])
    assert fb.dtype == np.float32 and np.all(np.abs(fb - g["frames"]) <= pose_tol(g["frames"]))
    np.testing.assert_allclose(mp.score_samples_batch(g["X"]), g["logp"], rtol=1e-9, atol=1e-7)


def test_spline_coeffs_are_mutable_like_the_reference(golden_case):
    """Callers overwrite spline.coeffs with aligned coefficients (motion_primitive_constraints.py:113);
    evaluation must follow the current array."""
    name, data, g = golden_case
    mp = _primitive(data)
    sp = mp.back_project(g["S"][0], False)
    sp.coeffs = np.array(g["coeffs"][1])
    scale = max(1.0, np.abs(g["frames"]).max())
    np.testing.assert_allclose(sp.get_motion_vector(), g["frames"][1], rtol=0, atol=4e-12 * scale)
    buffered = sp.get_buffered_motion_vector()
    assert sp.get_buffered_motion_vector() is buffered
    assert sp.get_domain() == (g["knots"][0], g["knots"][-1])
    assert sp.n_pose_parameters == g["coeffs"].shape[2]


def test_gaussian_mixture_surface(golden_case):
    name, data, g = golden_case
    gmm = _primitive(data).gaussian_mixture_model
    np.testing.assert_allclose(gmm.score_samples(g["X"]), g["logp"], rtol=1e-9, atol=1e-7)
    np.testing.assert_allclose(gmm.score(g["X"]), float(g["score_mean"]), rtol=1e-9, atol=1e-7)
    pc = g["precisions_cholesky"]
    np.testing.assert_allclose(gmm.precisions_cholesky_, pc, rtol=1e-9, atol=1e-9 * np.abs(pc).max())
    assert gmm.n_dims == g["X"].shape[1] and gmm.weights_.shape == (pc.shape[0],)
    with pytest.raises(ValueError):
        gmm.score(g["X"][0])              # 1-D input: sklearn raises ValueError too
    X, y = gmm.sample(500, device=True, seed=3)
    assert X.shape == (500, gmm.n_dims) and np.all(np.diff(y) >= 0)


def test_wrapper_format_dispatch_and_getters():
    data = synthetic.make_tiny_primitive(seed=1)                # translation_maxima [1,1,1]
    w_legacy, w_v3 = HipMotionPrimitiveModelWrapper(), HipMotionPrimitiveModelWrapper()
    w_legacy._initialize_from_json(None, data)
    w_v3._initialize_from_json(None, synthetic.to_mgrd_v3_json(data))
    assert not w_legacy.mgrd and not w_v3.mgrd
    s = np.array([0.3, -0.2, 0.9])
    fa = w_legacy.back_project(s, use_time_parameters=False).get_motion_vector()
    fb = w_v3.back_project(s, use_time_parameters=False).get_motion_vector()
    np.testing.assert_array_equal(fa, fb)
    assert w_v3.get_n_canonical_frames() == 12 and w_v3.get_n_spatial_components() == 3
    assert w_v3.get_n_time_components() == 0
    assert w_v3.get_spatial_eigen_vectors().shape == (3, 7 * 7)
    assert w_v3.sample_low_dimensional_vectors(5).shape == (5, 3)
    assert w_v3.sample_low_dimensional_vector().shape == (1, 3)
    assert w_v3.sample(False).get_motion_vector().shape == (12, 7)
    assert w_v3.back_project_time_function(s) == list(range(12))
    assert len(w_v3.get_animated_joints()) == 1
    assert w_v3.get_gaussian_mixture_model().score_samples(s.reshape(1, -1)).shape == (1,)
    static = HipMotionPrimitiveModelWrapper()
    static._initialize_from_json(None, {"name": "idle", "spatial_coeffs": np.zeros((4, 7)).tolist(),
                                        "knots": [0, 0, 0, 0, 3, 3, 3, 3], "n_canonical_frames": 4})
    assert static.get_n_spatial_components() == 1 and static.sample_low_dimensional_vector() == [0]


def test_semantic_label_and_time_parameter_errors():
    data = synthetic.make_tiny_primitive()
    data["semantic_label"] = {"left": 0, "right": 1}
    mp = _primitive(data)
    sp = mp.back_project(np.array([0.1, 0.2, 0.3, 1.0]), False)
    assert sp.semantic_annotation == "right" and len(sp.low_dimensional_parameters) == 3
    with pytest.raises(ValueError, match="Unknown semantic label"):
        mp.back_project(np.array([0.1, 0.2, 0.3, 7.0]), False)
    fresh = HipMotionPrimitive(None)
    with pytest.raises(AssertionError):
        fresh.sample_low_dimensional_vector()


def test_batched_candidate_scoring_matches_the_reference_loop():
    """evaluate_samples_using_constraints: per-sample loop of the reference vs one fused launch."""
    data = synthetic.make_walk_primitive(seed=0)
    mp = _primitive(data)
    cp = c_oracle.COraclePrimitive(data)
    np.random.seed(5)
    samples = mp.sample_low_dimensional_vector(300)             # n_random_samples = 300 preset
    cons = [{"type": "position", "t": 155.0, "weight": 1.0, "target": [40.0, None, -30.0]},
            {"type": "direction", "t": 155.0, "weight": 1.0, "target": [0.5, 1.0]}]
    nan = np.nan
    cons_c = np.array([[0, 155.0, 1.0, 40.0, nan, -30.0, 0, 0], [1, 155.0, 1.0, 0.5, 1.0, 0.0, 0.0, 1.0]])
    ref = cp.keyframe_errors_f64(samples, cons_c)
    np.testing.assert_allclose(HipSampleFilter.score_samples(mp, samples, cons), ref, rtol=1e-10, atol=1e-9)

    class Constraints(object):
        constraints, min_error, evaluations = cons, None, 0

    cobj = Constraints()
    best, err = evaluate_samples_using_constraints(samples, mp, cobj, None)
    best_idx, min_error = 0, np.inf
    for i, e in enumerate(ref):                                   # motion_primitive_generator.py:251-257
        if min_error > e:
            min_error, best_idx = e, i
    np.testing.assert_array_equal(best, samples[best_idx])
    assert abs(err - min_error) <= 1e-9 and cobj.evaluations == 300 and cobj.min_error == err
    # scoring through full frames agrees with the fused scorer (root position at the last canonical frame)
    frames = mp.back_project_frames_batch(samples, times=[155.0])
    d = np.sqrt((frames[:, 0, 0] - 40.0) ** 2 + (frames[:, 0, 2] + 30.0) ** 2)
    pos_only = HipSampleFilter.score_samples(mp, samples, cons[:1])
    np.testing.assert_allclose(d, pos_only, rtol=0, atol=2e-4)


def test_graph_of_primitives_option_evaluation():
    prims = synthetic.make_graph_primitives(4)
    pset = HipPrimitiveSet(prims)
    names = [p["name"] for p in prims]
    cons = {n: [{"type": "position", "t": float(p["n_canonical_frames"] - 1), "weight": 1.0, "target": [10.0, None, 5.0]}]
            for n, p in zip(names, prims)}
    best, results = pset.evaluate_options(names, cons, n_samples=256, rng_seed=11)
    assert best in names and set(results) == set(names)
    for n, p in zip(names, prims):
        cp = c_oracle.COraclePrimitive(p)
        np.random.seed(11)
        S = pset.nodes[n].sample_low_dimensional_vector(256)
        ref = cp.keyframe_errors_f64(S, np.array([[0, float(p["n_canonical_frames"] - 1), 1.0, 10.0, np.nan, 5.0, 0, 0]]))
        assert abs(results[n][1] - ref.min()) <= 1e-9
        np.testing.assert_array_equal(results[n][0], S[int(np.argmin(ref))])
    assert best == names[int(np.argmin([results[n][1] for n in names]))]


def test_graph_node_call_sites():
    """MotionStateGraphNode's hot-path call sites (motion_state_graph_node.py:183-272) on the HIP wrapper."""
    data = synthetic.make_walk_primitive(seed=0)
    cp = c_oracle.COraclePrimitive(data)
    node = HipMotionStateGraphNode()
    node.init_from_dict("walk", {"name": "leftStance", "mm": data})
    assert (node.action_name, node.name) == ("walk", "leftStance")
    rng = np.random.default_rng(3)
    S = rng.standard_normal((6, 40))
    ref_frames = cp.frames_f64(S)
    ref_len = arc_length_xz(ref_frames[:, :, :3])
    for i in range(3):
        assert abs(node.get_step_length_for_sample(S[i]) - ref_len[i]) <= 1e-9 * max(1.0, ref_len[i])
        d = np.linalg.norm(ref_frames[i, -1, :3] - ref_frames[i, 0, :3])
        assert abs(node.get_step_length_for_sample(S[i], "distance") - d) <= 1e-9 * max(1.0, d)
    np.testing.assert_allclose(node.get_step_lengths_for_samples(S), ref_len, rtol=2e-6)
    with pytest.raises(NotImplementedError):
        node.get_step_length_for_sample(S[0], "other")
    np.random.seed(9)
    node.update_motion_stats(n_samples=5)
    np.random.seed(9)
    lens = [arc_length_xz(cp.frames_f64(node.sample_low_dimensional_vector())[0][:, :3]) for _ in range(5)]
    assert abs(node.average_step_length - np.median(lens)) <= 1e-9 * np.median(lens)
    assert node.n_standard_transitions == 0
    assert node.predict_gmm(("walk", "rightStance"), S[0]) is node.get_gaussian_mixture_model()
    assert not node.has_transition_model(("walk", "rightStance"))
    assert node.search_best_sample(None, None) == (np.inf, None)
    cons = [{"type": "position", "t": 155.0, "weight": 1.0, "target": [20.0, None, 10.0]}]
    np.random.seed(4)
    err, best = node.search_best_sample_gpu(cons, 512)
    np.random.seed(4)
    S2 = node.sample_low_dimensional_vectors(512)
    ref = cp.keyframe_errors_f64(S2, np.array([[0, 155.0, 1.0, 20.0, np.nan, 10.0, 0, 0]]))
    assert abs(err - ref.min()) <= 1e-9 and np.array_equal(best, S2[int(np.argmin(ref))])


def test_batched_objective_functions_match_the_reference_formulas():
    """objective_functions.py:95-267 batched on the GPU against the oracle's line-by-line restatement: residual
    vectors, error sums, naturalness term, analytic mixture Jacobian, forward-difference kinematic Jacobian."""
    from morphablegraphs_amd import objective_functions as of
    from oracle import mg_oracle as orc
    data = synthetic.make_walk_primitive(seed=0)
    mp = _primitive(data)
    op = orc.OraclePrimitive(data)
    np.random.seed(11)
    S = mp.sample_low_dimensional_vector(40)
    L = S.shape[1]
    cons = [{"type": "position", "t": 155.0, "weight": 1.5, "target": [40.0, None, -30.0]},
            {"type": "position", "t": 77.0, "weight": 1.0, "target": [20.0, 0.0, -15.0]},
            {"type": "direction", "t": 155.0, "weight": 0.7, "target": [0.5, 1.0]}]

    class Constraints(object):
        constraints, min_error, evaluations = cons, None, 0

    ref_res = op.keyframe_residuals(S, cons)
    ref_lp = orc.gmm_log_prob(S, op.weights, op.means, op.prec_chol)
    error_scale, quality_scale, init_error_sum = 2.0, 0.01, 3.0
    c = Constraints()
    data6 = (mp, c, None, error_scale, quality_scale, init_error_sum)

    np.testing.assert_allclose(of.obj_spatial_error_sum(S, (mp, c, None)), ref_res.sum(axis=1), rtol=1e-10, atol=1e-9)
    assert c.evaluations == 40 and abs(c.min_error - ref_res[-1].sum()) <= 1e-9
    assert isinstance(of.obj_spatial_error_sum(S[3], (mp, c, None)), float)

    got = of.obj_spatial_error_sum_and_naturalness(S, data6)
    np.testing.assert_allclose(got, error_scale * ref_res.sum(axis=1) - ref_lp * quality_scale, rtol=1e-10, atol=1e-9)

    rv = of.obj_spatial_error_residual_vector(S, data6)
    assert rv.shape == (40, L)                                   # 3 residuals zero-padded to n_variables
    np.testing.assert_allclose(rv[:, :3], ref_res / init_error_sum, rtol=1e-10, atol=1e-9)
    assert not rv[:, 3:].any()
    rvn = of.obj_spatial_error_residual_vector_and_naturalness(S, data6)
    np.testing.assert_allclose(rvn[:, :3], (ref_res * error_scale - (ref_lp * quality_scale)[:, None]) / init_error_sum,
                               rtol=1e-10, atol=1e-9)
    assert not rvn[:, 3:].any() and of.obj_spatial_error_residual_vector_and_naturalness(S[0], data6).shape == (L,)

    jac = of.log_likelihood_jac(S, mp)
    np.testing.assert_allclose(jac, op.log_likelihood_jac(S), rtol=1e-8, atol=1e-9 * np.abs(jac).max())
    far = op.means[0] + 1e4                                       # exp(score) underflows: ones, like the reference
    np.testing.assert_array_equal(of.log_likelihood_jac(far, mp), np.ones(L))

    # kinematic Jacobian: approx_fprime's forward differences, all n*(L+1) points in one launch
    eps = 1e-7
    kin = of.spatial_error_jac(S[:4], (mp, c, None), eps)
    for b in range(4):
        f0 = op.keyframe_errors(S[b], cons)[0]
        for i in (0, 7, L - 1):
            sp = S[b].copy()
            sp[i] += eps
            fd = (op.keyframe_errors(sp, cons)[0] - f0) / eps
            assert abs(kin[b, i] - fd) <= 1e-4 * max(1.0, abs(fd)), (b, i, kin[b, i], fd)
    full = of.obj_spatial_error_sum_and_naturalness_jac(S[:4], data6, eps)
    np.testing.assert_allclose(full, jac[:4] * data6[-2] + kin * data6[-1], rtol=1e-12, atol=1e-12)
    with pytest.raises(ValueError):     # previous frames with a zero root quaternion have no heading to align to
        of.obj_spatial_error_sum(S, (mp, c, np.zeros((2, 79))))
    mp.close() if hasattr(mp, "close") else None


def test_generator_sampling_modes_and_batched_local_optimization():
    """MotionPrimitiveGenerator.generate_constrained_sample (motion_primitive_generator.py:126-162) on the HIP
    back end: gpu_batch / random_discrete pick the reference loop's first minimum; cluster_tree_search returns the
    exhaustive optimum of the tree's stored samples; the leastsq local optimization with the batched
    finite-difference Jacobian lands where scipy's own lmdif lands on the oracle's (CPU, float64) objective."""
    from scipy.optimize import leastsq
    from morphablegraphs_amd.motion_primitive_generator import (HipMotionPrimitiveGenerator, SAMPLING_MODE_CLUSTER_TREE_SEARCH,
                                                                 SAMPLING_MODE_RANDOM)
    from morphablegraphs_amd.candidate_scoring import SAMPLING_MODE_GPU_BATCH
    from oracle import mg_oracle as orc
    data = synthetic.make_tiny_primitive(seed=3)
    node = HipMotionStateGraphNode()
    node.init_from_dict("walk", {"name": "leftStance", "mm": data})
    op = orc.OraclePrimitive(data)
    t_end = float(op.n_canonical_frames - 1)
    cons = [{"type": "position", "t": t_end, "weight": 1.0, "target": [0.8, None, -0.4]},
            {"type": "position", "t": 0.5 * t_end, "weight": 1.0, "target": [0.4, None, -0.2]},
            {"type": "direction", "t": t_end, "weight": 0.05, "target": [0.2, 1.0]}]

    class Constraints(object):
        def __init__(self):
            self.constraints, self.min_error, self.evaluations = list(cons), None, 0
            self.motion_primitive_name, self.use_local_optimization = "leftStance", False

    cfg = {"n_random_samples": 200, "use_constraints": True, "use_transition_model": False, "use_local_coordinates": True,
           "constrained_sampling_mode": SAMPLING_MODE_GPU_BATCH, "n_cluster_search_candidates": 2,
           "local_optimization_settings": {"start_error_threshold": 0.0, "error_scale_factor": 1.0, "quality_scale_factor": 0.1,
                                           "method": "leastsq", "max_iterations": 500, "verbose": False}}
    gen = HipMotionPrimitiveGenerator({("walk", "leftStance"): node}, cfg, "walk")
    picks = {}
    for mode in (SAMPLING_MODE_GPU_BATCH, SAMPLING_MODE_RANDOM):
        cfg["constrained_sampling_mode"] = mode
        gen.set_algorithm_config(cfg)
        c = Constraints()
        np.random.seed(21)
        picks[mode] = gen.generate_constrained_sample(node, c)
        np.random.seed(21)
        samples = node.sample_low_dimensional_vectors(200)
        ref = op.keyframe_errors(samples, cons)
        best_idx, min_error = orc.first_min_argmin(ref)                  # the reference's loop
        np.testing.assert_array_equal(picks[mode], samples[best_idx])
        assert abs(c.min_error - min_error) <= 1e-9 and c.evaluations == 200
    np.testing.assert_array_equal(picks[SAMPLING_MODE_GPU_BATCH], picks[SAMPLING_MODE_RANDOM])

    # cluster tree: brute force over the stored samples == find_best_example_exhaustive
    class Tree(object):
        pass
    node.cluster_tree = Tree()
    np.random.seed(5)
    node.cluster_tree.data = node.sample_low_dimensional_vectors(500)
    cfg["constrained_sampling_mode"] = SAMPLING_MODE_CLUSTER_TREE_SEARCH
    gen.set_algorithm_config(cfg)
    c = Constraints()
    got = gen.generate_constrained_sample(node, c)
    ref = op.keyframe_errors(node.cluster_tree.data, cons)
    np.testing.assert_array_equal(got, node.cluster_tree.data[int(np.argmin(ref))])
    assert abs(c.min_error - ref.min()) <= 1e-9

    # local optimization: batched-Jacobian leastsq vs scipy's lmdif on the oracle objective
    cfg["constrained_sampling_mode"] = SAMPLING_MODE_GPU_BATCH
    gen.set_algorithm_config(cfg)
    c = Constraints()
    c.use_local_optimization = True
    np.random.seed(21)
    opt = gen.generate_constrained_sample(node, c)
    start = picks[SAMPLING_MODE_GPU_BATCH]
    L = start.shape[0]

    def cpu_objective(s, init_error_sum):   # obj_spatial_error_residual_vector_and_naturalness, objective_functions.py:239-267
        nll = -orc.gmm_log_prob(s[None], op.weights, op.means, op.prec_chol)[0] * 0.1
        r = list(op.keyframe_residuals(s, cons)[0] * 1.0 + nll)
        while len(r) < L:
            r.append(0.0)
        return np.array(r) / init_error_sum
    init = max(abs(np.sum(cpu_objective(start, 1.0))), 1.0)
    ref_opt = leastsq(cpu_objective, start, args=(init,), maxfev=500)[0]
    f_opt, f_ref, f_start = (np.sum(cpu_objective(x, init) ** 2) for x in (opt, ref_opt, start))
    assert f_opt < f_start
    assert abs(f_opt - f_ref) <= 1e-6 * max(1.0, f_ref), (f_opt, f_ref)
    np.testing.assert_allclose(opt, ref_opt, rtol=0, atol=1e-4 * max(1.0, np.abs(ref_opt).max()))
    # one launch per Jacobian instead of L + 1 objective calls
    assert gen.numerical_minimizer.n_launches < 200
    spline, params = gen.generate_constrained_motion_spline(c, None)
    assert spline.get_motion_vector().shape[1] == op.n_dim and params.shape == (L,)


def test_motion_state_graph_from_zip(tmp_path):
    """MotionStateGraphLoader._build_from_zip_file (reference motion_model/motion_state_graph_loader.py:184-242) on
    the HIP back end: every statistical primitive of the zip resident on the GPU, node types from the meta
    information, transitions and their types, the start node, cached vs recomputed step statistics, the stored
    samples of a cluster tree for the brute-force search."""
    from morphablegraphs_amd.motion_state_graph import HipMotionStateGraph, NODE_TYPE_START, NODE_TYPE_END
    prims = synthetic.make_graph_primitives(4)
    lists = [{k: (v.tolist() if hasattr(v, "tolist") else v) for k, v in p.items()} for p in prims]
    actions = {"walk": {"primitives": {"beginLeftStance": lists[0], "leftStance": synthetic.to_mgrd_v3_json(lists[1]),
                                       "endRightStance": lists[2]},
                        "info": {"start_states": ["beginLeftStance"], "end_states": ["endRightStance"]}},
               "pick": {"primitives": {"reach": lists[3]}, "info": {}}}
    transitions = {"walk_beginLeftStance": ["walk_leftStance"], "walk_leftStance": ["walk_leftStance", "walk_endRightStance"],
                   "walk_endRightStance": ["pick_reach"]}
    np.random.seed(9)
    stored = HipMotionPrimitive(None)
    stored._initialize_from_json(prims[1])
    tree_samples = stored.sample_low_dimensional_vector(64)
    path = str(tmp_path / "graph.zip")
    stats = {"walk": {"beginLeftStance": {"average_step_length": 12.5, "n_standard_transitions": 1}}}
    synthetic.write_graph_zip(path, actions, transitions=transitions, start_node=("walk", "walk_beginLeftStance"),
                              cluster_trees={("walk", "leftStance"): tree_samples}, stats=stats)
    graph = HipMotionStateGraph().load_from_zip(path)
    assert sorted(graph.nodes) == [("pick", "reach"), ("walk", "beginLeftStance"), ("walk", "endRightStance"), ("walk", "leftStance")]
    assert graph.start_node == ("walk", "beginLeftStance")                       # "walk_" prefix stripped, loader :236-240
    assert graph.nodes[("walk", "beginLeftStance")].node_type == NODE_TYPE_START
    assert graph.nodes[("walk", "endRightStance")].node_type == NODE_TYPE_END
    assert graph.nodes[("pick", "reach")].node_type == "single_primitive"
    edges = graph.nodes[("walk", "leftStance")].outgoing_edges
    assert edges[("walk", "leftStance")].transition_type == "standard" and edges[("walk", "endRightStance")].transition_type == "end"
    assert graph.nodes[("walk", "endRightStance")].outgoing_edges[("pick", "reach")].transition_type == "action_transition"
    # cached statistics are taken, missing ones are computed from GPU back-projections
    assert graph.nodes[("walk", "beginLeftStance")].average_step_length == 12.5
    node = graph.nodes[("walk", "leftStance")]
    assert node.average_step_length > 0 and node.n_standard_transitions == 1
    # every node back-projects through its own primitive; v3 and legacy files give the same model
    for (key, node), p in zip(sorted(graph.nodes.items()), [prims[3], prims[0], prims[2], prims[1]]):
        s = np.zeros(node.get_n_spatial_components())
        frames = node.back_project(s, use_time_parameters=False).get_motion_vector()
        ref = c_oracle.COraclePrimitive(p).frames_f64(s[None])[0]
        np.testing.assert_allclose(frames, ref, rtol=0, atol=1e-9 * max(1.0, np.abs(ref).max()))
    np.testing.assert_allclose(graph.nodes[("walk", "leftStance")].cluster_tree.data, tree_samples, rtol=0, atol=0)
    graph.close()


def test_baseline_config_sizes_graph_walk_step_and_optimizer_batch():
    """BASELINE.json configs[2] and [4] at their stated sizes, through size-independent checks: a graph-walk step over
    the 16 primitives of a MotionStateGraph with 4096 candidates per option (winner and error of every option against
    the C oracle's per-sample loop), and a 128k-candidate optimizer batch of the objective (a seeded subset against
    the oracle, the rest through the identity objective = error_scale * sum(residuals) - quality_scale * log p)."""
    from morphablegraphs_amd import objective_functions as of
    prims = synthetic.make_graph_primitives(16)
    pset = HipPrimitiveSet(prims)
    names = [p["name"] for p in prims]
    cons = {n: [{"type": "position", "t": float(p["n_canonical_frames"] - 1), "weight": 1.0, "target": [10.0, None, 5.0]},
                {"type": "direction", "t": 0.5 * float(p["n_canonical_frames"] - 1), "weight": 0.2, "target": [0.0, 1.0]}]
            for n, p in zip(names, prims)}
    best, results = pset.evaluate_options(names, cons, n_samples=4096, rng_seed=3)
    for n, p in zip(names, prims):
        cp = c_oracle.COraclePrimitive(p)
        np.random.seed(3)
        S = pset.nodes[n].sample_low_dimensional_vector(4096)
        t = float(p["n_canonical_frames"] - 1)
        ref = cp.keyframe_errors_f64(S, np.array([[0, t, 1.0, 10.0, np.nan, 5.0, 0, 0], [1, 0.5 * t, 0.2, 0.0, 1.0, 0.0, 0.0, 1.0]]))
        idx = int(np.argmin(ref))
        assert abs(results[n][1] - ref[idx]) <= 1e-9 * max(1.0, ref[idx])
        np.testing.assert_array_equal(results[n][0], S[idx])
    assert best == names[int(np.argmin([results[n][1] for n in names]))]

    data = synthetic.make_walk_primitive(seed=0)
    mp = _primitive(data)
    cp = c_oracle.COraclePrimitive(data)
    B = 128 * 1024
    rng = np.random.default_rng(17)
    S = (0.5 * rng.standard_normal((B, 40))).astype(np.float32)
    wcons = [{"type": "position", "t": 155.0, "weight": 1.0, "target": [40.0, None, -30.0]},
             {"type": "direction", "t": 155.0, "weight": 0.5, "target": [0.5, 1.0]}]
    obj = of.obj_spatial_error_sum_and_naturalness(S, (mp, wcons, None, 2.0, 0.1, 1.0))
    res = of.obj_spatial_error_residual_vector(S, (mp, wcons, None, 2.0, 0.1, 1.0))
    lp = mp._prim.gmm_log_prob(S, dtype=np.float64)
    assert obj.shape == (B,) and res.shape == (B, 40) and np.isfinite(obj).all()
    np.testing.assert_allclose(obj, 2.0 * res[:, :2].sum(axis=1) - 0.1 * lp, rtol=1e-12, atol=1e-9)
    idx = rng.choice(B, size=64, replace=False)
    ref = cp.keyframe_errors_f64(S[idx].astype(np.float64), np.array([[0, 155.0, 1.0, 40.0, np.nan, -30.0, 0, 0], [1, 155.0, 0.5, 0.5, 1.0, 0.0, 0.0, 1.0]]))
    np.testing.assert_allclose(res[idx, :2].sum(axis=1), ref, rtol=1e-10, atol=1e-9)
    np.testing.assert_allclose(lp[idx], cp.log_prob_f64(S[idx].astype(np.float64)), rtol=1e-10, atol=1e-8)


def test_device_sampled_candidates_stay_on_the_gpu():
    """gpu_batch with gpu_sampling: component counts from NumPy's stream, latents from the device Philox sampler,
    score + first-minimum argmin on the device, only the winner read back.  The winner must be a row the sampler
    produced for that seed, its error must be the minimum of the errors of all rows (oracle), and the generator
    must route to this path."""
    from morphablegraphs_amd.candidate_scoring import sample_and_evaluate_on_device, SAMPLING_MODE_GPU_BATCH
    from morphablegraphs_amd.motion_primitive_generator import HipMotionPrimitiveGenerator
    data = synthetic.make_walk_primitive(seed=0)
    node = HipMotionStateGraphNode()
    node.init_from_dict("walk", {"name": "leftStance", "mm": data})
    cp = c_oracle.COraclePrimitive(data)
    cons = [{"type": "position", "t": 155.0, "weight": 1.0, "target": [40.0, None, -30.0]},
            {"type": "direction", "t": 155.0, "weight": 1.0, "target": [0.5, 1.0]}]
    n = 4096
    np.random.seed(123)
    best, err = sample_and_evaluate_on_device(node, cons, n, seed=77)
    np.random.seed(123)
    counts = np.random.multinomial(n, np.asarray(data["gmm_weights"]) / np.sum(data["gmm_weights"]))
    X, comp = node.motion_primitive._prim.gmm_sample(counts, 77, dtype=np.float32)     # same seed, same rows
    ref = cp.keyframe_errors_f64(X.astype(np.float64), np.array([[0, 155.0, 1.0, 40.0, np.nan, -30.0, 0, 0], [1, 155.0, 1.0, 0.5, 1.0, 0.0, 0.0, 1.0]]))
    idx = int(np.argmin(ref))
    np.testing.assert_array_equal(best.astype(np.float32), X[idx])
    assert abs(err - ref[idx]) <= 1e-9 * max(1.0, ref[idx])

    class Constraints(object):
        def __init__(self):
            self.constraints, self.min_error, self.evaluations = list(cons), None, 0
            self.motion_primitive_name, self.use_local_optimization = "leftStance", False
    cfg = {"n_random_samples": 512, "constrained_sampling_mode": SAMPLING_MODE_GPU_BATCH, "gpu_sampling": True, "gpu_sampling_seed": 5,
           "local_optimization_settings": {"start_error_threshold": 0.0, "error_scale_factor": 1.0, "quality_scale_factor": 0.1,
                                           "method": "leastsq", "max_iterations": 50}}
    gen = HipMotionPrimitiveGenerator({("walk", "leftStance"): node}, cfg, "walk")
    c = Constraints()
    s1 = gen.generate_constrained_sample(node, c)
    assert s1.shape == (40,) and c.evaluations == 512 and np.isfinite(c.min_error)
    s2 = gen.generate_constrained_sample(node, c)
    assert not np.array_equal(s1, s2)                      # the seed advances per call


def test_graph_walk_step_on_device_with_one_stream_per_option():
    """evaluate_options_on_device: every option's sampler / scorer / argmin enqueued on its own stream before any
    result is read.  Same answer as the synchronous per-option path fed with the same device-sampled rows, and the
    winner is the option with the smallest error (np.argmin over options, graph_walk_planner.py:191-192)."""
    prims = synthetic.make_graph_primitives(6)
    names = [p["name"] for p in prims]
    cons = {n: [{"type": "position", "t": float(p["n_canonical_frames"] - 1), "weight": 1.0, "target": [10.0, None, 5.0]}]
            for n, p in zip(names, prims)}
    for separate in (False, True):
        pset = HipPrimitiveSet(prims, separate_streams=separate)
        np.random.seed(31)
        best, results = pset.evaluate_options_on_device(names, cons, n_samples=2048, seed=100)
        np.random.seed(31)
        for k, (n, p) in enumerate(zip(names, prims)):
            cp = c_oracle.COraclePrimitive(p)
            w = np.asarray(p["gmm_weights"], dtype=np.float64)
            counts = np.random.multinomial(2048, w / w.sum())
            X, _ = pset.nodes[n]._prim.gmm_sample(counts, 100 + k, dtype=np.float32)
            ref = cp.keyframe_errors_f64(X.astype(np.float64), np.array([[0, float(p["n_canonical_frames"] - 1), 1.0, 10.0, np.nan, 5.0, 0, 0]]))
            idx = int(np.argmin(ref))
            np.testing.assert_array_equal(results[n][0].astype(np.float32), X[idx])
            assert abs(results[n][1] - ref[idx]) <= 1e-9 * max(1.0, ref[idx])
        assert best == names[int(np.argmin([results[n][1] for n in names]))]
        # a second step reuses the per-option buffers
        best2, results2 = pset.evaluate_options_on_device(names, cons, n_samples=2048, seed=200)
        assert set(results2) == set(names)


def test_planner_step_notices_constraints_rewritten_in_place():
    """The planner step keeps the device constraint set of the previous step when an option's constraints compare equal value by
    value (no keys, no cache look-up: half of the step's host time): a target rewritten IN PLACE between two steps, and the shared
    set rewritten by another caller, must both be noticed."""
    from morphablegraphs_amd.candidate_scoring import cached_constraint_set
    prims = synthetic.make_graph_primitives(4)
    names = [p["name"] for p in prims]

    def constraints(x):
        return {n: [{"type": "position", "t": float(p["n_canonical_frames"] - 1), "weight": 1.0, "target": [x, None, 5.0]},
                    {"type": "direction", "t": float(p["n_canonical_frames"] - 1), "weight": 0.5, "target": [0.3, 1.0]}] for n, p in zip(names, prims)}

    def step(pset, cons):
        np.random.seed(3)
        best, res = pset.evaluate_options_on_device(names, cons, n_samples=1024, seed=7)
        return best, {n: (res[n][0].copy(), res[n][1]) for n in names}
    pset, fresh = HipPrimitiveSet(prims), HipPrimitiveSet(prims)
    cons = constraints(10.0)
    b1, r1 = step(pset, cons)
    b1b, r1b = step(pset, cons)                              # the remembered sets: the same answer
    for n in names:
        np.testing.assert_array_equal(r1[n][0], r1b[n][0])
        assert r1[n][1] == r1b[n][1]
    for n in names:
        cons[n][0]["target"][0] = -35.0                      # in place: same dicts, same lists
    b2, r2 = step(pset, cons)
    b2f, r2f = step(fresh, constraints(-35.0))               # a set that never saw the old target
    assert any(r2[n][1] != r1[n][1] for n in names)
    for n in names:
        np.testing.assert_array_equal(r2[n][0], r2f[n][0])
        assert r2[n][1] == r2f[n][1]
    assert b2 == b2f
    # somebody else scores the same kind of constraints on one of the primitives with other values: the shared set is rewritten
    other = constraints(77.0)[names[1]]
    cached_constraint_set(pset.nodes[names[1]]._prim, other, None, None)
    b3, r3 = step(pset, cons)
    for n in names:
        np.testing.assert_array_equal(r3[n][0], r2[n][0])
        assert r3[n][1] == r2[n][1]


def test_planner_step_notices_array_targets_rewritten_in_place():
    """targets held as NumPy arrays (and a fresh array of more than one element on the second step) must neither be
    compared by identity nor raise 'truth value of an array is ambiguous'; key names are part of the comparison."""
    prims = synthetic.make_graph_primitives(3)
    names = [p["name"] for p in prims]

    def constraints(x, as_array=True):
        wrap = (lambda v: np.array(v, dtype=np.float64)) if as_array else (lambda v: list(v))
        return {n: [{"type": "position", "t": float(p["n_canonical_frames"] - 1), "weight": 1.0, "target": wrap([x, np.nan, 5.0])},
                    {"type": "direction", "t": float(p["n_canonical_frames"] - 1), "weight": 0.5, "target": wrap([0.3, 1.0])}] for n, p in zip(names, prims)}

    def step(pset, cons):
        np.random.seed(3)
        best, res = pset.evaluate_options_on_device(names, cons, n_samples=1024, seed=7)
        return best, {n: (res[n][0].copy(), res[n][1]) for n in names}
    pset, fresh = HipPrimitiveSet(prims), HipPrimitiveSet(prims)
    cons = constraints(10.0)
    b1, r1 = step(pset, cons)
    b1b, r1b = step(pset, constraints(10.0))                 # fresh arrays with the same values: the same answer, no ValueError
    assert b1 == b1b and all(r1[n][1] == r1b[n][1] for n in names)
    for n in names:
        cons[n][0]["target"][0] = -35.0                      # the array rewritten in place
    b2, r2 = step(pset, cons)
    b2f, r2f = step(fresh, constraints(-35.0, as_array=False))
    assert any(r2[n][1] != r1[n][1] for n in names)
    for n in names:
        np.testing.assert_array_equal(r2[n][0], r2f[n][0])
        assert r2[n][1] == r2f[n][1]
    assert b2 == b2f


def _options_step_raw(pset, names, cons, n, seed, dtype, skeleton=None, prev_frames=None):
    """One evaluate_options_on_device step; returns per option (result record, all errors, all candidates) as the device left them."""
    np.random.seed(9)
    best, results = pset.evaluate_options_on_device(names, cons, n_samples=n, seed=seed, dtype=dtype, skeleton=skeleton, prev_frames=prev_frames)
    out = {}
    for name in names:
        prim = pset.nodes[name]._prim
        d_x, d_e, d_r = pset._buffers[(name, n, np.dtype(dtype).str)]
        L = prim.n_gmm_dims
        out[name] = (results[name][0].copy(), results[name][1], prim.ctx.download(d_e, (n,), np.float64), prim.ctx.download(d_x, (n, L), dtype))
    return best, out


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_planner_step_in_one_launch_is_bit_identical_to_the_per_option_chains(dtype):
    """mg_options_step as ONE launch (mg_options_fused_kernel: sample -> score -> first minimum -> winner copy, last
    workgroup per option reduces) against the chain of sampler / scorer / argmin launches per option
    (MG_OPT_OPTIONS_STEP = 1): the same candidates, the same errors, the same winners, bit for bit -- with root and
    forward-kinematics constraints, in global coordinates, for batch sizes whose component counts are not multiples of 16."""
    prims = synthetic.make_graph_primitives(7)
    names = [p["name"] for p in prims]
    joints, animated = synthetic.make_skeleton(19)
    hip_sk = _capi.Skeleton(joints, animated)
    cons = {}
    for n, p in zip(names, prims):
        tlast = float(p["n_canonical_frames"] - 1)
        cons[n] = [{"type": "position", "t": tlast, "weight": 1.0, "target": [10.0, None, 5.0]},
                   {"type": "direction", "t": (tlast / 2.0), "weight": 0.5, "target": [0.3, 1.0]},
                   {"type": "joint_position", "joint": "RightHand", "t": tlast, "weight": 2.0, "target": [12.0, 90.0, 4.0]}]
    pset = HipPrimitiveSet(prims)
    for n_samples in (2048, 777):
        pset.ctx.set_option(_capi.MG_OPT_OPTIONS_STEP, 1)
        b1, r1 = _options_step_raw(pset, names, cons, n_samples, 41, dtype, skeleton=hip_sk)
        pset.ctx.set_option(_capi.MG_OPT_OPTIONS_STEP, 0)
        pset.ctx.profile_enable(1)
        pset.ctx.profile_reset()
        b2, r2 = _options_step_raw(pset, names, cons, n_samples, 41, dtype, skeleton=hip_sk)
        assert pset.ctx.profile_get(7)[1] == 1                 # the whole step was ONE launch of the fused kernel ...
        assert pset.ctx.profile_get(4)[1] == 0 and pset.ctx.profile_get(2)[1] == 0   # ... no sampler, no scorer launch
        pset.ctx.profile_enable(0)
        assert b1 == b2
        for name in names:
            np.testing.assert_array_equal(r1[name][3].view(np.uint8), r2[name][3].view(np.uint8))   # candidates
            np.testing.assert_array_equal(r1[name][2].view(np.uint64), r2[name][2].view(np.uint64))  # errors
            np.testing.assert_array_equal(r1[name][0].view(np.uint64), r2[name][0].view(np.uint64))  # winning latent
            assert r1[name][1] == r2[name][1]
            assert r1[name][1] == np.min(r1[name][2])


def test_planner_step_at_the_size_of_configs2_returns_what_its_buffers_hold():
    """BASELINE configs[2] at its stated size -- 16 primitives x 4096 device-sampled candidates each -- through the one-launch step,
    several steps in a row (counts drawn ahead from the second on): every option's record is the first minimum of the errors the
    step left on the device and that row of its candidates, the counts sum to the batch, the best option is the smallest error."""
    prims = synthetic.make_graph_primitives(16)
    names = [p["name"] for p in prims]
    cons = {nm: [{"type": "position", "t": float(p["n_canonical_frames"] - 1), "weight": 1.0, "target": [10.0, None, 5.0]},
                 {"type": "direction", "t": float(p["n_canonical_frames"] - 1), "weight": 1.0, "target": [0.5, 1.0]}] for nm, p in zip(names, prims)}
    pset = HipPrimitiveSet(prims, separate_streams=False)
    n = 4096
    for step in range(4):
        best, res = pset.evaluate_options_on_device(names, cons, n, seed=70 + step, device_counts=True)
        errs = []
        for nm in names:
            prim = pset.nodes[nm]._prim
            d_x, d_e, d_r = pset._buffers[(nm, n, np.dtype(np.float32).str)]
            e = prim.ctx.download(d_e, (n,), np.float64)
            x = prim.ctx.download(d_x, (n, prim.n_gmm_dims), np.float32)
            w = int(np.argmin(e))
            assert res[nm][1] == e[w], (step, nm)
            np.testing.assert_array_equal(np.asarray(res[nm][0], dtype=np.float64), x[w].astype(np.float64))
            assert pset.last_counts[nm].sum() == n and np.all(pset.last_counts[nm] >= 0)
            errs.append(e[w])
        assert best == names[int(np.argmin(errs))]


def test_planner_steps_publish_complete_records_fenced_or_not():
    """The one-launch planner step publishes its partials and its host records WITHOUT memory fences (csrc/mg_options.hip, "THE
    PUBLICATION PROTOCOL'S CONTRACT": write-through stores + vmcnt + barrier + counter).  500 steps of BASELINE configs[2]'s shape
    (16 options x 4096, counts drawn on the device a step ahead): after EVERY step every option's record equals the first minimum of
    the errors the step left on the device and that row of its candidates -- a torn or stale record, or a winner reduced from
    partials that had not arrived, fails here.  The option count changes between steps (16, 8, 16, 5, ...) so that the records'
    layout in the pinned block moves (completion flags used to sit behind the records, where an earlier step's bytes
    could read as this step's sequence number).  Once more with the release / acquire form of the same kernel
    (MG_OPT_OPTIONS_STEP 3): identical records."""
    prims = synthetic.make_graph_primitives(16)
    names = [p["name"] for p in prims]
    cons = {nm: [{"type": "position", "t": float(p["n_canonical_frames"] - 1), "weight": 1.0, "target": [10.0, None, 5.0]},
                 {"type": "direction", "t": float(p["n_canonical_frames"] - 1), "weight": 1.0, "target": [0.5, 1.0]}] for nm, p in zip(names, prims)}
    pset = HipPrimitiveSet(prims, separate_streams=False)
    n = 4096
    subsets = [names, names[:8], names, names[3:8], names[:12]]

    def run(steps, check_every):
        out = []
        for step in range(steps):
            opts = subsets[step % len(subsets)] if step % 7 == 6 else names
            best, res = pset.evaluate_options_on_device(opts, {nm: cons[nm] for nm in opts}, n, seed=9000 + step, device_counts=True)
            out.append((best, [(nm, res[nm][1], np.asarray(res[nm][0]).tobytes()) for nm in opts]))
            if step % check_every:
                continue
            errs = []
            for nm in opts:
                prim = pset.nodes[nm]._prim
                d_x, d_e, d_r = pset._buffers[(nm, n, np.dtype(np.float32).str)]
                e = prim.ctx.download(d_e, (n,), np.float64)
                w = int(np.argmin(e))
                assert res[nm][1] == e[w], (step, nm, res[nm][1], e[w])
                x = prim.ctx.download(d_x.ptr.value + w * prim.n_gmm_dims * 4, (prim.n_gmm_dims,), np.float32)
                np.testing.assert_array_equal(np.asarray(res[nm][0], dtype=np.float64), x.astype(np.float64), err_msg="step %d %s" % (step, nm))
                errs.append(e[w])
            assert best == opts[int(np.argmin(errs))]
        return out
    free = run(500, 1)
    pset.ctx.set_option(_capi.MG_OPT_OPTIONS_STEP, 3)
    try:
        fenced = run(60, 1)
    finally:
        pset.ctx.set_option(_capi.MG_OPT_OPTIONS_STEP, 0)
    assert fenced == free[:60]


def test_planner_step_with_component_counts_drawn_on_the_device(monkeypatch):
    """mg_options_step_device_counts: the counts are the ones the oracle's restatement of the device draw gives (Philox4x32-10
    keyed by seed + option index, histogram of n categorical draws), bit for bit; given those counts the step is the step
    mg_options_step makes (same candidates, errors, winners); the draw is reproducible, sums to n, and is distributed like
    numpy.random.multinomial's counts (means and variances over many seeds)."""
    from oracle import mg_oracle as orc
    prims = synthetic.make_graph_primitives(7)
    names = [p["name"] for p in prims]
    cons = {}
    for n, p in zip(names, prims):
        tlast = float(p["n_canonical_frames"] - 1)
        cons[n] = [{"type": "position", "t": tlast, "weight": 1.0, "target": [10.0, None, 5.0]},
                   {"type": "direction", "t": (tlast / 2.0), "weight": 0.5, "target": [0.3, 1.0]}]
    pset = HipPrimitiveSet(prims)
    for n_samples, seed in ((2048, 41), (777, 5)):
        best, res = pset.evaluate_options_on_device(names, cons, n_samples=n_samples, seed=seed, device_counts=True)
        counts = {k: v.copy() for k, v in pset.last_counts.items()}
        for k, (name, p) in enumerate(zip(names, prims)):
            ref = orc.device_multinomial_counts(n_samples, p["gmm_weights"], seed + k)
            np.testing.assert_array_equal(counts[name], ref, err_msg=name)
            assert counts[name].sum() == n_samples
        best_b, res_b = pset.evaluate_options_on_device(names, cons, n_samples=n_samples, seed=seed, device_counts=True)
        assert best_b == best and all(res_b[nm][1] == res[nm][1] for nm in names)          # reproducible
        # the host route fed the same counts: the same step
        feed = iter([counts[nm] for nm in names])
        monkeypatch.setattr(np.random, "multinomial", lambda n, pvals: next(feed))
        best_h, res_h = pset.evaluate_options_on_device(names, cons, n_samples=n_samples, seed=seed)
        monkeypatch.undo()
        assert best_h == best
        for nm in names:
            np.testing.assert_array_equal(res_h[nm][0].view(np.uint64), res[nm][0].view(np.uint64))
            assert res_h[nm][1] == res[nm][1]
    # counts drawn AHEAD: a step's kernel draws the counts of seed + 1 as well, the next step with those seeds starts without a
    # counts kernel -- the same counts, candidates and winners as with the kernel in front (MG_OPT_OPTIONS_STEP 2)
    runs = {}
    for mode in (0, 2):
        pset.ctx.set_option(_capi.MG_OPT_OPTIONS_STEP, mode)
        runs[mode] = []
        for seed in (300, 301, 302, 310, 311):
            best, res = pset.evaluate_options_on_device(names, cons, n_samples=1500, seed=seed, device_counts=True)
            runs[mode].append((best, {nm: (res[nm][0].copy(), res[nm][1]) for nm in names}, {k: v.copy() for k, v in pset.last_counts.items()}))
            for k, (name, p) in enumerate(zip(names, prims)):
                np.testing.assert_array_equal(pset.last_counts[name], orc.device_multinomial_counts(1500, p["gmm_weights"], seed + k), err_msg="%s seed %d mode %d" % (name, seed, mode))
    pset.ctx.set_option(_capi.MG_OPT_OPTIONS_STEP, 0)
    for (b0, r0, c0), (b2, r2, c2) in zip(runs[0], runs[2]):
        assert b0 == b2
        for nm in names:
            np.testing.assert_array_equal(r0[nm][0].view(np.uint64), r2[nm][0].view(np.uint64))
            assert r0[nm][1] == r2[nm][1]
    # distribution: one option, many seeds
    name, p = names[0], prims[0]
    w = np.asarray(p["gmm_weights"], dtype=np.float64)
    w = w / w.sum()
    n_samples, reps = 4096, 300
    draws = []
    for s in range(reps):
        pset.evaluate_options_on_device([name], {name: cons[name]}, n_samples=n_samples, seed=1000 + s, device_counts=True)
        draws.append(pset.last_counts[name].copy())
    draws = np.array(draws, dtype=np.float64)
    mean, var = draws.mean(axis=0), draws.var(axis=0)
    sd = np.sqrt(n_samples * w * (1 - w))
    assert np.all(np.abs(mean - n_samples * w) <= 5 * sd / np.sqrt(reps) + 1e-9), (mean, n_samples * w)
    if len(w) > 1:
        assert np.all(np.abs(var - sd ** 2) <= 0.35 * sd ** 2 + 1.0), (var, sd ** 2)


def test_planner_step_falls_back_to_one_stream_for_large_mixtures():
    """More than 16 mixture components (prefix sums staged in the context's scratch buffer) or the VALU sampler forced: the
    options of a step share that ONE staging buffer, so mg_options_step must not run them on side streams.
    The step must equal mg_option_step called option by option on the context's own stream."""
    prims = [synthetic.make_primitive(seed=300 + i, n_components=24, n_frames=60, n_gmm=17, name="big%d" % i) for i in range(5)]
    names = [p["name"] for p in prims]
    cons = {n: [{"type": "position", "t": 59.0, "weight": 1.0, "target": [3.0, None, -2.0]}] for n in names}
    pset = HipPrimitiveSet(prims)
    for force_valu in (0, 1):
        pset.ctx.set_option(_capi.MG_OPT_FORCE_VALU_SAMPLE, force_valu)
        _, step = _options_step_raw(pset, names, cons, 1500, 7, np.float32)
        np.random.seed(9)
        for k, (name, p) in enumerate(zip(names, prims)):
            prim = pset.nodes[name]._prim
            w = np.asarray(p["gmm_weights"], dtype=np.float64)
            counts = np.random.multinomial(1500, w / w.sum())
            X, _ = prim.gmm_sample(counts, 7 + k, dtype=np.float32)
            np.testing.assert_array_equal(step[name][3], X)
            cp = c_oracle.COraclePrimitive(p)
            ref = cp.keyframe_errors_f64(X.astype(np.float64), np.array([[0, 59.0, 1.0, 3.0, np.nan, -2.0, 0, 0]]))
            np.testing.assert_allclose(step[name][2], ref, rtol=1e-9, atol=1e-9)
            assert step[name][1] == np.min(step[name][2])
    pset.ctx.set_option(_capi.MG_OPT_FORCE_VALU_SAMPLE, 0)


def test_scoring_in_global_coordinates_through_the_adaptors():
    """prev_frames given and constraints not `is_local`: evaluate_samples_using_constraints, the objective
    functions, the generator without use_local_coordinates and the planner's option step all align every candidate
    to the last previous frame first (reference motion_primitive_constraints.py:110-114; the planner never
    localises, graph_walk_planner.py:179).  Checked against the oracle that transforms the control points, and
    against the reference's other route to the same number: localising the constraint with the inverse aligning
    transform (transform_constraints_to_local_cos, motion_primitive_constraints.py:268-291) and scoring locally."""
    from oracle import mg_oracle as orc
    from morphablegraphs_amd import objective_functions as of
    from morphablegraphs_amd.motion_primitive_generator import HipMotionPrimitiveGenerator
    joints, animated = synthetic.make_skeleton()
    hip_sk = _capi.Skeleton(joints, animated)
    data = synthetic.make_walk_primitive(seed=0)
    node = HipMotionStateGraphNode()
    node.init_from_dict("walk", {"name": "leftStance", "mm": data})
    op = orc.OraclePrimitive(data)
    rng = np.random.default_rng(21)
    prev_frames = op.back_project_frames(rng.standard_normal(40)).copy()
    prev_frames[:, 0] += 300.0
    prev_frames[:, 2] -= 150.0
    cons = [{"type": "position", "t": 155.0, "weight": 1.0, "target": [340.0, None, -120.0]},
            {"type": "direction", "t": 155.0, "weight": 0.2, "target": [1.0, 0.3]},
            {"type": "joint_position", "joint": "RightHand", "t": 77.5, "weight": 0.5, "target": [320.0, 100.0, -140.0]}]

    class RefSkeleton(object):
        aligning_root_node, aligning_root_dir, root = "Hips", (0.0, 0.0, 1.0), "Hips"

    class Constraints(object):
        def __init__(self, clist, is_local=False):
            self.constraints, self.min_error, self.evaluations = list(clist), None, 0
            self.motion_primitive_name, self.use_local_optimization = "leftStance", False
            self.is_local, self.skeleton, self.hip_skeleton, self.start_pose = is_local, RefSkeleton(), hip_sk, None

    S = rng.standard_normal((300, 40))
    ref = op.aligned_residuals(S, cons, prev_frames[-1], joints, animated, "Hips")
    c = Constraints(cons)
    best, err = evaluate_samples_using_constraints(S, node, c, prev_frames)
    idx = int(np.argmin(ref.sum(axis=1)))
    np.testing.assert_array_equal(best, S[idx])
    assert abs(err - ref.sum(axis=1)[idx]) < 1e-7 and c.evaluations == 300 and c.min_error == err
    np.testing.assert_allclose(of.obj_spatial_error_sum(S[:40], (node, c, prev_frames)), ref[:40].sum(axis=1), rtol=1e-9, atol=1e-8)
    np.testing.assert_allclose(of.obj_spatial_error_residual_vector(S[3], (node, c, prev_frames, 1.0, 1.0, 2.0))[:3], ref[3] / 2.0, rtol=1e-9, atol=1e-8)
    # is_local constraints are never aligned, previous frames or not
    loc = Constraints(cons, is_local=True)
    np.testing.assert_allclose(of.obj_spatial_error_sum(S[:10], (node, loc, prev_frames)),
                               of.obj_spatial_error_sum(S[:10], (node, loc, None)), rtol=0, atol=0)
    # the other route: one candidate, its own aligning transform, the position target taken into its local frame
    for b in (0, 7):
        coeffs = op.back_project_spatial_coeffs(S[b])
        aligned = orc.align_coeffs_to_previous_frame(coeffs, prev_frames[-1], joints, animated, "Hips")
        # recover the 4x4 transform from two control points (rotation about y + xz translation)
        d0, d1 = coeffs[8][[0, 2]] - coeffs[0][[0, 2]], aligned[8][[0, 2]] - aligned[0][[0, 2]]
        phi = np.arctan2(d0[1], d0[0]) - np.arctan2(d1[1], d1[0])
        m = np.eye(4)
        m[0, 0], m[0, 2], m[2, 0], m[2, 2] = np.cos(phi), np.sin(phi), -np.sin(phi), np.cos(phi)
        m[[0, 2], 3] = aligned[0][[0, 2]] - (m[:3, :3] @ coeffs[0][:3])[[0, 2]]
        tgt = np.linalg.inv(m) @ np.array([340.0, 0.0, -120.0, 1.0])
        local = [{"type": "position", "t": 155.0, "weight": 1.0, "target": [tgt[0], None, tgt[2]]}]
        np.testing.assert_allclose(of.obj_spatial_error_sum(S[b], (node, local, None)), ref[b, 0], rtol=1e-8, atol=1e-7)
    # the generator in global coordinates
    cfg = {"n_random_samples": 256, "constrained_sampling_mode": "random_discrete", "use_local_coordinates": False,
           "local_optimization_settings": {"start_error_threshold": 0.0, "error_scale_factor": 1.0, "quality_scale_factor": 0.1,
                                           "method": "leastsq", "max_iterations": 50}}
    gen = HipMotionPrimitiveGenerator({("walk", "leftStance"): node}, cfg, "walk")
    c2 = Constraints(cons)
    np.random.seed(9)
    s = gen.generate_constrained_sample(node, c2, prev_frames=prev_frames)
    np.random.seed(9)
    X = node.sample_low_dimensional_vectors(256)
    r2 = op.aligned_residuals(X, cons, prev_frames[-1], joints, animated, "Hips").sum(axis=1)
    np.testing.assert_array_equal(s, X[int(np.argmin(r2))])
    assert abs(c2.min_error - r2.min()) < 1e-7
    # the planner's option step, aligned
    pset = HipPrimitiveSet([data])
    np.random.seed(4)
    best_name, results = pset.evaluate_options_on_device([data["name"]], {data["name"]: Constraints(cons)}, 1024, seed=50,
                                                         prev_frames=prev_frames)
    np.random.seed(4)
    w = np.asarray(data["gmm_weights"], dtype=np.float64)
    Xd, _ = pset.nodes[data["name"]]._prim.gmm_sample(np.random.multinomial(1024, w / w.sum()), 50, dtype=np.float32)
    r3 = op.aligned_residuals(Xd.astype(np.float64), cons, prev_frames[-1], joints, animated, "Hips").sum(axis=1)
    np.testing.assert_array_equal(results[best_name][0].astype(np.float32), Xd[int(np.argmin(r3))])
    assert abs(results[best_name][1] - r3.min()) < 1e-7


def test_reference_shaped_hand_constraints_through_the_objectives():
    """TwoHandConstraint and a GlobalTransformConstraint with position and orientation, handed over as the
    reference's objects: the error sum and the residual vector keep the reference's entries (three for the two-hand
    constraint, two_hand_constraint.py:66-74; ONE for position + orientation, global_transform_constraint.py:70-77)."""
    from oracle import mg_oracle as orc
    from morphablegraphs_amd import objective_functions as of
    joints, animated = synthetic.make_skeleton()
    hip_sk = _capi.Skeleton(joints, animated)
    data = synthetic.make_walk_primitive(seed=0)
    node = HipMotionStateGraphNode()
    node.init_from_dict("walk", {"name": "leftStance", "mm": data})
    op = orc.OraclePrimitive(data)

    class RefSkeleton(object):
        root, aligning_root_node, aligning_root_dir = "Hips", "Hips", (0.0, 0.0, 1.0)

    class TwoHand(object):
        canonical_keyframe, weight_factor, skeleton = 100, 0.5, RefSkeleton()
        positions, orientations, joint_names = [[30.0, 95.0, 10.0], [-20.0, 99.0, 14.0]], [None, None], ["LeftHand", "RightHand"]

    class HeadPose(object):
        canonical_keyframe, weight_factor, skeleton, joint_name = 40, 2.0, RefSkeleton(), "Head"
        position, orientation = [0.0, 160.0, None], [0.9, 0.1, -0.3, 0.2]

    class Constraints(object):
        def __init__(self):
            self.constraints, self.min_error, self.evaluations = [TwoHand(), HeadPose()], None, 0
            self.is_local, self.skeleton, self.hip_skeleton, self.start_pose = True, RefSkeleton(), hip_sk, None

    S = np.random.default_rng(2).standard_normal((12, 40))
    expect = np.zeros((12, 4))
    for b in range(12):
        coeffs = op.back_project_spatial_coeffs(S[b])
        f1 = orc.spline_frames(op.knots, coeffs, [100.0])[0]
        f2 = orc.spline_frames(op.knots, coeffs, [40.0])[0]
        expect[b, :3] = 0.5 * np.asarray(orc.two_hand_residuals(f1, joints, animated, TwoHand.joint_names, TwoHand.positions))
        expect[b, 3] = 2.0 * (orc.point_distance(HeadPose.position, orc.joint_global_position(f2, joints, animated, "Head")) +
                              orc.joint_orientation_error(f2, joints, animated, "Head", HeadPose.orientation))
    c = Constraints()
    np.testing.assert_allclose(of.obj_spatial_error_sum(S, (node, c, None)), expect.sum(axis=1), rtol=1e-9, atol=1e-8)
    res = of.obj_spatial_error_residual_vector(S, (node, c, None, 1.0, 1.0, 1.0))
    assert res.shape == (12, 40)                                        # zero padded to n_variables columns
    np.testing.assert_allclose(res[:, :4], expect, rtol=1e-9, atol=1e-8)
    assert not res[:, 4:].any()
    best, err = evaluate_samples_using_constraints(S, node, c)
    np.testing.assert_array_equal(best, S[int(np.argmin(expect.sum(axis=1)))])


def test_keyframe_classes_against_the_references_own_classes():
    """tests/golden/keyframe_classes.npz: TwoHandConstraintSet and FeetConstraint run by the reference itself on its own motion
    splines of the walk model (forward kinematics: the oracle's).  Objects with those classes' attributes go through the
    objectives; the entries are the reference's get_residual_vector_spline values times the weight factor, as
    MotionPrimitiveConstraints.get_residual_vector (motion_primitive_constraints.py:140-146) and .evaluate (:118-121) apply it."""
    from conftest import load_golden
    from morphablegraphs_amd import objective_functions as of
    g = load_golden("keyframe_classes")
    joints, animated = synthetic.make_skeleton()
    hip_sk = _capi.Skeleton(joints, animated)
    data = synthetic.make_walk_primitive(seed=0)
    node = HipMotionStateGraphNode()
    node.init_from_dict("walk", {"name": "leftStance", "mm": data})
    S = np.ascontiguousarray(g["S"])

    class RefSkeleton(object):
        root, aligning_root_node, aligning_root_dir = "Hips", "Hips", (0.0, 0.0, 1.0)

    class Obj(object):
        def __init__(self, **kw):
            self.__dict__.update(kw)

    cons, expect = [], []
    for ci in range(int(g["n_two_hand"])):
        w = float(g["two_hand_weight_%d" % ci])
        cons.append(Obj(canonical_keyframe=int(g["two_hand_keyframe_%d" % ci]), weight_factor=w, skeleton=RefSkeleton(),
                        positions=[p for p in g["two_hand_positions_%d" % ci]], orientations=[None, None],
                        joint_names=[str(n) for n in g["two_hand_joints_%d" % ci]]))
        expect.append(w * g["two_hand_residuals_%d" % ci])
    for ci in range(int(g["n_feet"])):
        w = float(g["feet_weight_%d" % ci])
        cons.append(Obj(canonical_keyframe=int(g["feet_keyframe_%d" % ci]), weight_factor=w, skeleton=RefSkeleton(),
                        left=g["feet_left_%d" % ci], right=g["feet_right_%d" % ci]))
        expect.append(w * g["feet_residuals_spline_%d" % ci])
    expect = np.hstack(expect)
    assert expect.shape == (len(S), 3 * int(g["n_two_hand"]) + int(g["n_feet"]))
    c = Obj(constraints=cons, min_error=None, evaluations=0, is_local=True, skeleton=RefSkeleton(), hip_skeleton=hip_sk, start_pose=None)
    res = of.obj_spatial_error_residual_vector(S, (node, c, None, 1.0, 1.0, 1.0))
    np.testing.assert_allclose(res[:, :expect.shape[1]], expect, rtol=1e-9, atol=1e-8)
    assert not res[:, expect.shape[1]:].any()
    total = sum(float(g["two_hand_weight_%d" % ci]) * g["two_hand_error_%d" % ci] for ci in range(int(g["n_two_hand"]))) + \
        sum(float(g["feet_weight_%d" % ci]) * g["feet_error_%d" % ci] for ci in range(int(g["n_feet"])))
    np.testing.assert_allclose(of.obj_spatial_error_sum(S, (node, c, None)), total, rtol=1e-9, atol=1e-8)
    best, err = evaluate_samples_using_constraints(S, node, c)
    np.testing.assert_array_equal(best, S[int(np.argmin(total))])


def test_cluster_tree_training_data_in_batches():
    """The data-producing half of ClusterTreeBuilder (reference construction/cluster_tree_builder.py:159-192,
    293-301): threshold filter, best-of-2n by the reference's heap slice, back projection at the integer canonical
    frames -- each as one batched call, compared with the reference's per-sample loops run on the oracle."""
    import heapq
    from oracle import mg_oracle as orc
    from morphablegraphs_amd.cluster_tree_sampling import HipClusterTreeSampler
    data = synthetic.make_walk_primitive(seed=0)
    node = HipMotionStateGraphNode()
    node.init_from_dict("walk", {"name": "leftStance", "mm": data})
    op = orc.OraclePrimitive(data)
    sampler = HipClusterTreeSampler(n_samples=300)

    np.random.seed(5)
    X = node.sample_low_dimensional_vectors(300)
    ll = op.score_samples(X)
    thr = float(np.median(ll))
    np.random.seed(5)
    got = sampler._get_samples_using_threshold(node, threshold=thr, max_iter_count=5)
    np.random.seed(5)
    expect, count, it = [], 0, 0
    while count < 300 and it < 5:                               # the reference's loop, scores from the oracle
        Xi = node.sample_low_dimensional_vectors(300)
        for s, l in zip(Xi, op.score_samples(Xi)):
            if l > thr:
                expect.append(s)
                count += 1
        it += 1
    assert it < 5 and got.shape == (len(expect), 40)
    np.testing.assert_array_equal(got, np.asarray(expect))
    assert sampler._get_samples_using_threshold(node, threshold=1e9, max_iter_count=2) is None

    np.random.seed(6)
    best = sampler._get_best_samples(node)
    np.random.seed(6)
    X2 = node.sample_low_dimensional_vectors(600)
    heap = []
    for idx, l in enumerate(op.score_samples(X2)):
        heapq.heappush(heap, (-l, idx))
    ref = X2[[i for _, i in heap[:300]]]
    np.random.shuffle(ref)
    np.testing.assert_array_equal(best, ref)

    motions = sampler._back_project(node, X[:20])
    assert motions.shape == (20, 156, 79) and motions.dtype == np.float64
    ref_m = op.back_project_frames_batch(X[:20], time_points=np.arange(156.0))
    np.testing.assert_allclose(motions, ref_m, rtol=1e-11, atol=1e-11 * max(1.0, np.abs(ref_m).max()))
    np.testing.assert_array_equal(sampler._extract_features(node, X), X[:, :40])
    assert sampler.sample_data(node).shape == (300, 40)


def test_graph_primitives_share_one_device_arena(tmp_path):
    """A graph's primitives are bump-allocated from shared arena blocks (one device arena per
    graph); results are those of separately allocated primitives, bit for bit, and a block is released when its
    last primitive goes."""
    prims = synthetic.make_graph_primitives(5)
    ctx = _capi.Context(0)
    assert ctx.arena_bytes() == (0, 0)
    ctx.arena_begin(8 << 20)
    inside = [_capi.Primitive(ctx, p) for p in prims]
    ctx.arena_end()
    reserved, used = ctx.arena_bytes()
    assert reserved >= used > 0 and reserved % (8 << 20) == 0
    outside = [_capi.Primitive(ctx, p) for p in prims]
    assert ctx.arena_bytes() == (reserved, used)                       # closed: later primitives allocate on their own
    rng = np.random.default_rng(0)
    for a, b in zip(inside, outside):
        S = rng.standard_normal((40, a.n_components)).astype(np.float32)
        np.testing.assert_array_equal(a.back_project_frames(S).view(np.uint32), b.back_project_frames(S).view(np.uint32))
        np.testing.assert_array_equal(a.gmm_log_prob(S), b.gmm_log_prob(S))
        cs_a = _capi.ConstraintSet(a, [{"type": "position", "t": 3.0, "weight": 1.0, "target": [1.0, None, 2.0]}])
        cs_b = _capi.ConstraintSet(b, [{"type": "position", "t": 3.0, "weight": 1.0, "target": [1.0, None, 2.0]}])
        np.testing.assert_array_equal(a.score_constraints(cs_a, S), b.score_constraints(cs_b, S))
        cs_a.close()
        cs_b.close()
    for p in inside + outside:
        p.close()
    assert ctx.arena_bytes() == (0, 0)                                  # every block went with its last primitive
    ctx.close()
    # the graph loader uses it
    from morphablegraphs_amd.motion_state_graph import HipMotionStateGraph
    path = str(tmp_path / "graph.zip")
    lists = [{k: (v.tolist() if hasattr(v, "tolist") else v) for k, v in p.items()} for p in prims[:3]]
    synthetic.write_graph_zip(path, {"walk": {"primitives": {"a": lists[0], "b": lists[1], "c": lists[2]}, "info": {}}})
    gctx = _capi.Context(0)
    graph = HipMotionStateGraph(context=gctx).load_from_zip(path)
    assert gctx.arena_bytes()[1] > 0 and len(graph.nodes) == 3
    graph.close()
    assert gctx.arena_bytes() == (0, 0)
    gctx.close()


def test_the_ctypes_stub_in_integration_md_runs_as_printed():
    """INTEGRATION.md section 2 shows the binding a maintainer of the reference would add.  The code block is taken
    from the document as it stands, pointed at the in-tree library, and must reproduce the adaptor's results."""
    import os
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    text = open(os.path.join(root, "INTEGRATION.md")).read()
    block = re.search(r"```python\n(# morphablegraphs/motion_model/hip_backend.py.*?)```", text, re.S).group(1)
    lib = os.path.join(root, "morphablegraphs_amd", "csrc", "libmg_hip.so")
    assert 'C.CDLL("libmg_hip.so")' in block
    ns = {}
    exec(compile(block.replace('C.CDLL("libmg_hip.so")', "C.CDLL(%r)" % lib), "INTEGRATION.md", "exec"), ns)
    data = synthetic.make_walk_primitive(seed=0)
    backend = ns["HipBackend"](data)
    S = np.random.default_rng(0).standard_normal((20, 40)).astype(np.float32)
    prim = HipMotionPrimitive(None)
    prim._initialize_from_json(data)
    np.testing.assert_array_equal(backend.frames(S).view(np.uint32), prim.back_project_frames_batch(S).view(np.uint32))
    np.testing.assert_array_equal(backend.score_samples(S), prim.score_samples_batch(S.astype(np.float64)))


def test_model_with_time_latents_end_to_end():
    """A model whose mixture spans spatial AND time latents, as the reference's constructor writes it
    (motion_model_constructor.py:424): loads, samples at full width on sklearn's stream, scores all columns,
    back-projects from the spatial ones, and evaluates the time warp -- against vectors made by the reference."""
    from conftest import golden_model
    from morphablegraphs_amd.motion_primitive_wrapper import HipMotionPrimitiveModelWrapper
    data, g = golden_model("time_model")
    mp = HipMotionPrimitive(None)
    mp._initialize_from_json(data)
    n_s, n_t = int(g["n_spatial_components"]), int(g["n_time_components"])
    assert mp.has_time_parameters and (mp.get_n_spatial_components(), mp.get_n_time_components()) == (n_s, n_t)
    assert mp.gaussian_mixture_model.n_dims == n_s + n_t and mp._prim.n_gmm_dims == n_s + n_t
    np.random.seed(int(g["seed"]))
    S = mp.sample_low_dimensional_vector(len(g["S"]))
    np.testing.assert_allclose(S, g["S"], rtol=1e-12, atol=1e-12)
    np.testing.assert_allclose(mp.gaussian_mixture_model.score_samples(g["S"]), g["logp_S"], rtol=1e-9, atol=1e-7)
    np.testing.assert_allclose(mp.gaussian_mixture_model.precisions_cholesky_, g["precisions_cholesky"], rtol=1e-9, atol=1e-9)
    scale = max(1.0, np.abs(g["frames"]).max())
    for s, fr, ctf in zip(g["S"], g["frames"], g["canonical_time_functions"]):
        spline = mp.back_project(s, use_time_parameters=False)
        np.testing.assert_allclose(spline.get_motion_vector(), fr, rtol=0, atol=4e-12 * scale)
        np.testing.assert_array_equal(spline.low_dimensional_parameters, s)
        np.testing.assert_allclose(mp._back_transform_gamma_to_canonical_time_function(s[n_s:]), ctf, rtol=1e-12, atol=1e-11)
    np.testing.assert_allclose(mp.back_transform_gamma_to_canonical_time_function_batch(g["S"][:, n_s:]), g["canonical_time_functions"],
                               rtol=1e-12, atol=1e-11)
    np.testing.assert_allclose(mp._mean_temporal(), g["mean_temporal"], rtol=0, atol=1e-13)
    # float32 batch path reads the spatial columns of full-width rows
    fb = mp.back_project_frames_batch(g["S"])
    assert np.all(np.abs(fb - g["frames"]) <= 1e-5 + 2.0 ** -24 * np.abs(g["frames"]))
    # the time warp itself (use_time_parameters=True): the reference raises TypeError there on every NumPy >= 1.18;
    # the restatement with int(num) gives a monotone map from sample time to canonical time with the reference's end points
    warped = mp.back_project(g["S"][0], use_time_parameters=True)
    tf = np.asarray(warped.time_function)
    assert tf[0] == 0 and tf[-1] == mp.n_canonical_frames - 1 and np.all(np.diff(tf[1:-1]) > 0)
    assert warped.get_motion_vector().shape == (len(tf), mp.s_pca["n_dim"])
    # the wrapper's time function of a sample = the canonical time function of its time latents (wrapper :233-249)
    w = HipMotionPrimitiveModelWrapper()
    w._initialize_from_json(None, data)
    np.testing.assert_allclose(w.back_project_time_function(g["S"][2]), g["canonical_time_functions"][2], rtol=1e-12, atol=1e-11)
    np.testing.assert_allclose(w.back_project_time_functions(g["S"]), g["canonical_time_functions"], rtol=1e-12, atol=1e-11)
    # device sampler and the winner of a device-side option step come back at full width
    X, comp = mp.gaussian_mixture_model.sample(4096, device=True, seed=5)
    assert X.shape == (4096, n_s + n_t) and np.isfinite(X).all()
    w0 = np.asarray(data["gmm_weights"])
    mu = (w0[:, None] * np.asarray(data["gmm_means"])).sum(axis=0)
    assert np.abs(X.mean(axis=0) - mu).max() < 0.15
    from morphablegraphs_amd.candidate_scoring import sample_and_evaluate_on_device
    cons = [{"type": "position", "t": float(mp.n_canonical_frames - 1), "weight": 1.0, "target": [10.0, None, 5.0]}]
    best, err = sample_and_evaluate_on_device(mp, cons, 512, 3)
    assert best.shape == (n_s + n_t,) and np.isfinite(err)
    res = mp._prim.score_constraints(_capi.ConstraintSet(mp._prim, cons), best.reshape(1, -1))
    assert abs(res[0] - err) < 1e-4 * max(1.0, abs(err))


def test_time_warped_synthesis_of_a_batch_against_the_reference():
    """back_project(s, use_time_parameters=True).get_motion_vector() for a batch on the device: the spline's time
    functions (mg_time_function_sample: canonical time function, not-a-knot inversion, sampling) and the frames at every
    candidate's own times (mg_back_project_frames_at), against vectors the REFERENCE's own lines produced
    (tests/golden/time_model.npz: oracle/gen_golden.py runs motion_primitive.py:206-234,268-319 with np.linspace truncating its
    float sample count as NumPy <= 1.17 did), at speed 1 and 1.6, for float64 and float32 output; and against the host route."""
    from conftest import golden_model
    data, g = golden_model("time_model")
    mp = HipMotionPrimitive(None)
    mp._initialize_from_json(data)
    F, n_s = mp.n_canonical_frames, int(g["n_spatial_components"])
    S = g["S"]
    for tag, speed in (("speed10", 1.0), ("speed16", 1.6)):
        want_t, want_f, want_n = g["sample_time_functions_" + tag], g["warped_frames_" + tag], g["warped_lengths_" + tag]
        frames, lens, times = mp.back_project_warped_batch(S, speed)
        np.testing.assert_array_equal(lens, want_n)                    # the sample counts: exact
        scale = max(1.0, np.nanmax(np.abs(want_f)))
        for b in range(len(S)):
            n = int(lens[b])
            np.testing.assert_allclose(times[b, :n], want_t[b, :n], rtol=0, atol=1e-11 * F)        # FITPACK's B-spline form vs second derivatives
            assert times[b, 0] == 0.0 and times[b, n - 1] == F - 1 and np.all(np.isnan(times[b, n:]))
            np.testing.assert_allclose(frames[b, :n], want_f[b, :n], rtol=0, atol=4e-9 * scale)    # 1e-11 F in time x the frames' slope
            assert np.all(np.isnan(frames[b, n:]))
            # given the reference's times the frames are the float64 kernels' frames: 4e-12 scale
            exact = mp._prim.back_project_frames_at(S[b:b + 1, :n_s], want_t[b:b + 1, :n], None)
            np.testing.assert_allclose(exact[0], want_f[b, :n], rtol=0, atol=4e-12 * scale)
            grid = mp._prim.time_grid(want_t[b, :n])
            np.testing.assert_array_equal(exact[0], mp._prim.back_project_frames_f64(S[b:b + 1, :n_s], grid)[0])     # the same arithmetic
            grid.close()
            # the adaptor's single-sample calls
            np.testing.assert_array_equal(mp.back_project_time_function(S[b, n_s:], speed), times[b, :n])
            host = mp._invert_canonical_to_sample_time_function(g["canonical_time_functions"][b], speed)
            np.testing.assert_allclose(times[b, :n], host, rtol=0, atol=1e-11 * F)
        f32, _, _ = mp.back_project_warped_batch(S, speed, dtype=np.float32)
        ok = ~np.isnan(frames)
        np.testing.assert_array_equal(f32[ok], frames[ok].astype(np.float32))
    warped = mp.back_project(S[0], use_time_parameters=True)
    np.testing.assert_allclose(warped.get_motion_vector(), g["warped_frames_speed10"][0, :g["warped_lengths_speed10"][0]], rtol=0, atol=4e-9 * scale)
    # a larger batch of mixture draws: monotone maps with the reference's end points, every row of its own length
    np.random.seed(3)
    big = mp.sample_low_dimensional_vector(3000)
    frames, lens, times = mp.back_project_warped_batch(big, 1.0)
    assert lens.min() >= 2 and len(set(lens.tolist())) > 3 and frames.shape == (3000, lens.max(), mp.s_pca["n_dim"])
    for b in (0, 1, 777, 2999):
        n = int(lens[b])
        tf = times[b, :n]
        assert tf[0] == 0 and tf[-1] == F - 1 and np.all(np.diff(tf[1:-1]) > 0)
        op_t, op_f = None, None
    # rows too short for a candidate's samples are reported, not overrun
    t2, l2 = mp._prim.time_function_sample(big[:4, n_s:], 1.0, t_cap=8)
    assert np.all(l2 < 0) and np.all(-l2 == lens[:4])


_TRAJ_TOL = 2.0e-6     # the closest-point search's stated tolerance (tests/test_gpu_closest_point.py): the forward difference's noise


def _path_following_model():
    return synthetic.make_path_following_primitive(seed=0)


def test_trajectory_constraint_on_the_root_path():
    """TrajectoryConstraint.get_residual_vector / evaluate_motion_spline for the root joint (reference
    trajectory_constraint.py:79-121) in one launch per candidate batch.  Round 5: the device's search IS the reference's (scipy's
    L-BFGS-B restated for one variable; pinned by tests/test_gpu_closest_point.py against the reference's own vectors): here against
    the reference's call restated through scipy (orc.trajectory_residuals) and against the restatement the device mirrors
    (orc.closest_point), both ways within the search's stated tolerance (2e-6 of the distance's scale: the forward difference's
    noise); the monotone walk of rounds 2-4 (MG_OPT_TRAJECTORY_SEARCH 1) against ITS restatement at 1e-9."""
    from oracle import mg_oracle as orc
    data = _path_following_model()
    mp = _primitive(data)
    op = orc.OraclePrimitive(data)
    prim = mp._prim
    rng = np.random.default_rng(8)
    S = rng.standard_normal((37, 40))
    # a trajectory in the neighbourhood of the candidates' root paths: through points of one candidate's own path
    path0 = op.back_project_frames(S[0])[:, :3]
    cps = path0[::26].copy()
    cps[:, 0] += np.linspace(0.0, 6.0, len(cps))
    cps[:, 2] -= np.linspace(0.0, 4.0, len(cps))
    traj = _capi.Trajectory(prim, cps, granularity=1000)
    err, res = prim.score_trajectory(traj, S, min_u=0.0, weight=1.5, residuals=True)
    assert res.shape == (37, 156) and err.shape == (37,)
    np.testing.assert_allclose(err, res.mean(axis=1), rtol=1e-13)
    for b in (0, 5, 36):
        path = op.back_project_frames(S[b])[:, :3]
        min_u, chain = 0.0, []
        for p in path:
            pt, min_u = orc.closest_point(cps, p, min_u)
            chain.append(np.linalg.norm(p - pt))
        scale = max(1.0, 1.5 * max(chain))
        assert np.abs(res[b] - 1.5 * np.array(chain)).max() <= _TRAJ_TOL * scale
        ref = 1.5 * orc.trajectory_residuals(path, cps, 0.0)                 # the reference's own call (scipy), as its own chain
        assert np.abs(res[b] - ref).max() <= _TRAJ_TOL * scale, np.abs(res[b] - ref).max()
    prim.ctx.set_option(_capi.MG_OPT_TRAJECTORY_SEARCH, 1)
    try:
        err_w, res_w = prim.score_trajectory(traj, S, min_u=0.0, weight=1.5, residuals=True)
        for b in (0, 5, 36):
            path = op.back_project_frames(S[b])[:, :3]
            min_u, walk = 0.0, []
            for p in path:
                pt, min_u = orc.closest_point_walk(cps, p, min_u)
                walk.append(np.linalg.norm(p - pt))
            np.testing.assert_allclose(res_w[b], 1.5 * np.array(walk), rtol=1e-9, atol=1e-9)
        # (on these smooth forward paths the two searches agree: one basin ahead of the bound)
        assert np.abs(res_w - res).max() <= 1e-5 * max(1.0, res.max())
    finally:
        prim.ctx.set_option(_capi.MG_OPT_TRAJECTORY_SEARCH, 0)
    # a later start on the trajectory, float32 latents, another time grid (every second frame)
    err2 = prim.score_trajectory(traj, S.astype(np.float32), min_u=0.4)
    path = op.back_project_frames(S[3].astype(np.float32).astype(np.float64))[:, :3]
    np.testing.assert_allclose(err2[3], orc.trajectory_residuals(path, cps, 0.4).mean(), rtol=1e-5)
    grid = prim.time_grid(np.arange(0.0, 156.0, 2.0))
    e3, r3 = prim.score_trajectory(traj, S[:4], residuals=True, grid=grid)
    assert r3.shape == (4, 78)
    grid.close()
    # global coordinates: every candidate aligned to a previous frame (root as aligning node) or to a start pose first
    prev = op.back_project_frames(rng.standard_normal(40))[-1].copy()
    prev[:3] = [30.0, 88.0, -15.0]
    joints, animated = synthetic.make_skeleton()
    al = _capi.Skeleton(joints, animated).alignment_to(prev, 0)
    e_al, r_al = prim.score_trajectory(traj, S[:6], alignment={"joint": 0, "position": al["position"], "heading": al["heading"]}, residuals=True)
    for b in range(6):
        coeffs = orc.align_coeffs_to_previous_frame(op.back_project_spatial_coeffs(S[b]), prev, joints, animated, "Hips")
        path = orc.spline_frames(op.knots, coeffs, op.canonical_time_function())[:, :3]
        min_u, walk = 0.0, []
        for p in path:
            pt, min_u = orc.closest_point(cps, p, min_u)
            walk.append(np.linalg.norm(p - pt))
        np.testing.assert_allclose(r_al[b], walk, rtol=0, atol=_TRAJ_TOL * max(1.0, max(walk)))
    from morphablegraphs_amd.candidate_scoring import alignment_from_start_pose
    sp = {"position": [3.0, 2.0, 1.0], "orientation": [0.0, 25.0, 0.0]}
    e_sp, r_sp = prim.score_trajectory(traj, S[:3], alignment=alignment_from_start_pose(sp), residuals=True)
    for b in range(3):
        coeffs = orc.align_coeffs_to_start_pose(op.back_project_spatial_coeffs(S[b]), {"position": [3.0, 2.0, 1.0], "orientation": [0.0, 25.0, 0.0]})
        path = orc.spline_frames(op.knots, coeffs, op.canonical_time_function())[:, :3]
        min_u, walk = 0.0, []
        for p in path:
            pt, min_u = orc.closest_point(cps, p, min_u)
            walk.append(np.linalg.norm(p - pt))
        np.testing.assert_allclose(r_sp[b], walk, rtol=0, atol=_TRAJ_TOL * max(1.0, max(walk)))
    traj.close()


def test_trajectory_constraint_known_answer_on_a_straight_line():
    """A target spline through collinear, equally spaced control points is the straight line through them (Catmull-Rom reproduces
    linear data), so the closest point is the orthogonal projection, clamped to the segment and to the monotone-parameter rule:
    the distances have a closed form.  Root paths: parallel to the line at a known offset (distance = the offset); running ahead
    of the line's end (distance to the end point); running BACKWARDS along it (the parameter may not go back: distance to the point
    reached so far)."""
    from oracle import mg_oracle as orc
    data = _path_following_model()
    NB, D = 31, 79
    mean = np.array(data["mean_spatial_vector"]).reshape(NB, D)
    eig = np.zeros((40, NB, D))
    # the root path is the spline of the root control points: straight control points give a straight path x = 10 + 100 s, y = 90, z = 5
    s_ctrl = np.linspace(0.0, 1.0, NB)
    mean[:, 0], mean[:, 1], mean[:, 2] = 10.0 + 100.0 * s_ctrl, 90.0, 5.0
    eig[0, :, 1] = 1.0                       # latent 0 lifts the whole path by s[0] in y
    eig[1, :, 0] = -100.0 * s_ctrl           # latent 1 = 2 reverses the direction of travel: x = 10 + 100 s (1 - s[1])
    data["mean_spatial_vector"] = mean.reshape(-1).tolist()
    data["eigen_vectors_spatial"] = eig.reshape(40, -1).tolist()
    mp = _primitive(data)
    prim = mp._prim
    op = orc.OraclePrimitive(data)
    cps = np.stack([np.linspace(0.0, 120.0, 7), np.full(7, 90.0), np.full(7, 5.0)], axis=1)     # the line y = 90, z = 5, x in [0, 120]
    traj = _capi.Trajectory(prim, cps, granularity=1000)
    S = np.zeros((4, 40))
    S[1, 0] = 3.0                            # 3 above the line all the way
    S[2, 1] = -0.5                           # x = 10 + 150 s: past the end of the line from s = 11 / 15 on
    S[3, 1] = 2.0                            # x = 10 - 100 s: backwards
    # (the monotone walk: exact on a line.  The reference's search minimises the NORM, which has a kink where the point is on the
    # line, and stops at its tolerances: it is held to the reference's vectors instead, tests/test_gpu_closest_point.py)
    prim.ctx.set_option(_capi.MG_OPT_TRAJECTORY_SEARCH, 1)
    try:
        err, res = prim.score_trajectory(traj, S, min_u=0.0, weight=1.0, residuals=True)
    finally:
        prim.ctx.set_option(_capi.MG_OPT_TRAJECTORY_SEARCH, 0)
    paths = [op.back_project_frames(s)[:, :3] for s in S]
    np.testing.assert_allclose(res[0], 0.0, atol=1e-9)
    np.testing.assert_allclose(res[1], 3.0, rtol=1e-12, atol=1e-9)
    np.testing.assert_allclose(res[2], np.maximum(paths[2][:, 0] - 120.0, 0.0), atol=1e-9)
    reached = np.maximum.accumulate(np.clip(paths[3][:, 0], 0.0, 120.0))                        # the parameter never goes back
    np.testing.assert_allclose(res[3], np.abs(paths[3][:, 0] - reached), atol=1e-9)
    np.testing.assert_allclose(err, res.mean(axis=1), rtol=1e-13, atol=1e-12)
    traj.close()


def test_candidate_loop_with_a_trajectory_constraint():
    """evaluate_samples_using_constraints / HipSampleFilter / the device-sampling step with a trajectory constraint next to
    keyframe constraints (the path-following mix): errors add up per candidate before the first-minimum argmin; a
    reference-shaped TrajectoryConstraint object is converted (control points unpadded, min_u = min_arc_length / full)."""
    from oracle import mg_oracle as orc
    from morphablegraphs_amd.candidate_scoring import sample_and_evaluate_on_device, clear_constraint_cache
    data = _path_following_model()
    mp = _primitive(data)
    op = orc.OraclePrimitive(data)
    rng = np.random.default_rng(9)
    S = rng.standard_normal((64, 40))
    cps = op.back_project_frames(S[7])[::31, :3].copy()

    class _Spline(object):                                  # CatmullRomSpline's attributes (catmull_rom_spline.py:47-71)
        _catmullrom_basematrix = None
        control_points = [list(cps[0])] + [list(p) for p in cps] + [list(cps[-1]), list(cps[-1])]

    class _Trajectory(object):                              # TrajectoryConstraint's (trajectory_constraint.py:33-52)
        constraint_type = "trajectory"
        joint_name = "Hips"
        spline = _Spline()
        weight_factor = 0.7
        granularity = 1000
        full_arc_length = orc.catmull_rom_full_arc_length(cps)
        min_arc_length = 0.1 * full_arc_length
        is_collision_avoidance_constraint = False
    keyframe = {"type": "position", "t": 155.0, "weight": 1.0, "target": [float(cps[-1][0]), None, float(cps[-1][2])]}
    cons = [keyframe, _Trajectory()]
    errors = HipSampleFilter.score_samples(mp, S, cons)
    kf = op.keyframe_errors(S, [keyframe])
    for b in (0, 7, 63):
        tr = 0.7 * orc.trajectory_residuals(op.back_project_frames(S[b])[:, :3], cps, 0.1).mean()
        assert abs(errors[b] - (kf[b] + tr)) <= 2e-3 * max(1.0, kf[b] + tr)
    best, err = evaluate_samples_using_constraints(S, mp, cons)
    assert np.array_equal(best, S[int(np.argmin(errors))]) and abs(err - errors.min()) < 1e-9
    b2, e2 = sample_and_evaluate_on_device(mp, cons, 256, 11)
    assert b2.shape == (40,) and np.isfinite(e2) and e2 <= np.median(errors)
    _Trajectory.joint_name = "LeftHand"
    _Trajectory.skeleton = type("_Sk", (), {"root": "Hips"})()
    with pytest.raises(NotImplementedError):
        HipSampleFilter.score_samples(mp, S, cons)
    clear_constraint_cache()


def test_objective_functions_with_a_trajectory_constraint():
    """The batched objectives with a trajectory constraint in the list: obj_spatial_error_sum adds its weighted AVERAGE
    distance (MotionPrimitiveConstraints.evaluate -> evaluate_motion_spline, trajectory_constraint.py:79-86), the
    residual-vector objectives get its per-frame weighted distances as further columns
    (get_residual_vector_spline, :88-115) after the keyframe columns, zero padding and init_error_sum as for keyframes."""
    from oracle import mg_oracle as orc
    from morphablegraphs_amd import objective_functions as of
    from morphablegraphs_amd.candidate_scoring import clear_constraint_cache
    data = _path_following_model()
    mp = _primitive(data)
    op = orc.OraclePrimitive(data)
    rng = np.random.default_rng(10)
    S = rng.standard_normal((9, 40))
    cps = op.back_project_frames(S[2])[::31, :3].copy()
    keyframe = {"type": "position", "t": 155.0, "weight": 2.0, "target": [float(cps[-1][0]), None, float(cps[-1][2])]}
    traj = {"type": "trajectory", "control_points": cps, "min_u": 0.0, "weight": 0.5}
    class Constraints(object):                               # MotionPrimitiveConstraints' attributes the objectives touch
        constraints = [keyframe, traj]
        is_local = True
        min_error = 0.0
        evaluations = 0
    cons = Constraints()
    kf = op.keyframe_errors(S, [keyframe])
    walks = []
    for b in range(len(S)):
        min_u, walk = 0.0, []
        for p in op.back_project_frames(S[b])[:, :3]:
            pt, min_u = orc.closest_point(cps, p, min_u)
            walk.append(np.linalg.norm(p - pt))
        walks.append(0.5 * np.array(walk))
    walks = np.array(walks)
    err = of.obj_spatial_error_sum(S, (mp, cons, None))
    np.testing.assert_allclose(err, kf + walks.mean(axis=1), rtol=_TRAJ_TOL)
    assert cons.evaluations == len(S) and abs(cons.min_error - err[-1]) < 1e-12
    res = of.obj_spatial_error_residual_vector(S, (mp, cons, None, 1.0, 1.0, 4.0))
    assert res.shape == (9, 1 + 156)
    np.testing.assert_allclose(res[:, 0], kf / 4.0, rtol=1e-8)
    np.testing.assert_allclose(res[:, 1:], walks / 4.0, rtol=0, atol=_TRAJ_TOL * max(1.0, walks.max()))
    one = of.obj_spatial_error_residual_vector(S[4], (mp, cons, None, 1.0, 1.0, 4.0))
    np.testing.assert_array_equal(one, res[4])
    nat = of.obj_spatial_error_residual_vector_and_naturalness(S, (mp, cons, None, 3.0, 0.25, 2.0))
    nll = -mp.gaussian_mixture_model.score_samples(S) * 0.25
    np.testing.assert_allclose(nat[:, 1:], (walks * 3.0 + nll[:, None]) / 2.0, rtol=0, atol=_TRAJ_TOL * max(1.0, (walks * 3.0 + nll[:, None]).max()))
    jac = of.spatial_error_jac(S[:2], (mp, cons, None))
    assert jac.shape == (2, 40) and np.all(np.isfinite(jac))
    clear_constraint_cache()


def test_euclidean_feature_map_of_the_cluster_tree_builder():
    """The feature map under FeatureClusterTree construction (reference cluster_tree_builder.py:266-301 ->
    space_partitioning/features.py:125-153): every sample back-projected at the integer canonical frames, the global
    positions of a set of joints in every frame (one forward-kinematics launch instead of n x F x J Python calls), PCA on
    the host.  Positions against the rotation-matrix oracle (FK is anim_utils': PARITY UNPINNED); the `step` quirk of the
    reference (frames between multiples of step repeat the last computed cloud) reproduced."""
    from oracle import mg_oracle as orc
    from morphablegraphs_amd.cluster_tree_sampling import HipClusterTreeSampler
    joints, animated = synthetic.make_skeleton()
    sk = _capi.Skeleton(joints, animated)
    data = synthetic.make_walk_primitive(seed=0)
    w = HipMotionPrimitiveModelWrapper()
    w._initialize_from_json(None, data)
    op = orc.OraclePrimitive(data)
    sampler = HipClusterTreeSampler(n_samples=24)
    np.random.seed(3)
    S = sampler.sample_data(w)
    motions = sampler._back_project(w, S)
    assert motions.shape == (24, 156, 79)
    names = ["LeftUpLeg", "RightHand", "LeftHand_EndSite", "RightFoot", "Hips"]
    for step in (1, 4):
        clouds = sampler.map_motions_to_euclidean_space(w, motions, sk, names, step=step)
        assert clouds.shape == (24, 156 * len(names) * 3)
        for b, f in ((0, 0), (5, 77), (23, 155), (7, 6)):
            src = (f // step) * step                                   # the frame whose cloud the reference appends at f
            frame = orc.spline_frames(op.knots, op.back_project_spatial_coeffs(S[b][:40]), [float(src)])[0]
            expect = np.concatenate([orc.joint_global_position(frame, joints, animated, nm) for nm in names])
            got = clouds[b].reshape(156, -1)[f]
            np.testing.assert_allclose(got, expect, rtol=1e-9, atol=1e-8, err_msg="step %d sample %d frame %d" % (step, b, f))
    feats = sampler._extract_features(w, S, feature_type="euclidean_pca", skeleton=sk, joint_names=names, step=1)
    assert feats.shape[0] == 24 and 1 <= feats.shape[1] <= 24 and np.isfinite(feats).all()
    np.testing.assert_array_equal(sampler._extract_features(w, S), S[:, :40])
    # the raw entry point: known answer for the identity pose (every quaternion (1, 0, 0, 0)): offsets add up along the chain
    frame = np.zeros((1, 79))
    frame[0, :3] = [1.0, 2.0, 3.0]
    frame[0, 3::4] = 1.0
    pos = w.motion_primitive._prim.ctx.joint_positions(sk, ["LeftHand_EndSite", "Hips"], frame)
    chain = sk.chain("LeftHand_EndSite")
    np.testing.assert_allclose(pos[0, 0], [1.0, 2.0, 3.0] + sk.offsets[chain[1:]].sum(axis=0), atol=1e-12)
    np.testing.assert_allclose(pos[0, 1], [1.0, 2.0, 3.0], atol=1e-12)
