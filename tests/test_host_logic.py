"""Host-side logic that needs no GPU: the C-ABI library loads and exports every symbol the header
declares, sharding / argmin helpers, the sklearn-compatible sampler, format readers."""
import io
import json
import os
import re
import zipfile

import numpy as np
import pytest

from morphablegraphs_amd import _capi, distributed, model_io, synthetic
from morphablegraphs_amd.candidate_scoring import constraints_to_device_form
from morphablegraphs_amd.gaussian_mixture import sample_like_sklearn
from morphablegraphs_amd.motion_primitive_wrapper import mgrd_json_to_legacy

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_loads_and_exports_every_declared_symbol():
    header = open(os.path.join(ROOT, "include", "mg_hip.h")).read()
    declared = set(re.findall(r"\b(mg_[a-z0-9_]+)\s*\(", header))
    declared -= {"mg_primitive_desc", "mg_keyframe_constraint"}
    assert declared, "no declarations parsed"
    assert declared == set(_capi.EXPORTED_SYMBOLS), declared ^ set(_capi.EXPORTED_SYMBOLS)
    lib = _capi.load_library()
    missing = [s for s in sorted(declared) if not hasattr(lib, s)]
    assert not missing, missing
    assert lib.mg_version().startswith(b"mg_hip")
    assert lib.mg_status_string(-5) == b"covariance not positive definite"


def test_no_gpu_means_loud_failure_not_fallback():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(_capi.MGError) as ei:
        _capi.Context(0)
    assert ei.value.status == -2


def test_product_package_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "morphablegraphs_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, re.M), f
                assert "libmg_oracle" not in src, f


def test_shard_range_covers_everything_once():
    for n in (0, 1, 7, 8, 4096, 65536, 65537):
        for w in (1, 2, 3, 8):
            blocks = [distributed.shard_range(n, r, w) for r in range(w)]
            assert blocks[0][0] == 0 and blocks[-1][1] == n
            assert all(blocks[i][1] == blocks[i + 1][0] for i in range(w - 1))
            sizes = [e - b for b, e in blocks]
            assert max(sizes) - min(sizes) <= 1


def test_first_min_argmin_matches_reference_rule():
    f = distributed.first_min_argmin
    assert f([3.0, 1.0, 1.0, 2.0]) == (1, 1.0)
    assert f([np.nan, 2.0, np.nan, 2.0]) == (1, 2.0)
    assert f([]) == (0, float("inf"))
    assert f([np.inf, np.inf]) == (0, float("inf"))
    assert f([np.nan]) == (0, float("inf"))
    assert f([-np.inf, 0.0]) == (0, float("-inf"))


def test_sample_like_sklearn_is_bit_compatible():
    from sklearn.mixture import GaussianMixture
    from sklearn.mixture._gaussian_mixture import _compute_precision_cholesky
    data = synthetic.make_primitive(seed=3, n_components=6, n_frames=30, n_dim=11, n_gmm=3, dirichlet_weights=True)
    w, m, c = np.array(data["gmm_weights"]), np.array(data["gmm_means"]), np.array(data["gmm_covars"])
    gmm = GaussianMixture(n_components=3, covariance_type="full")
    gmm.weights_, gmm.means_, gmm.covariances_ = w, m, c
    gmm.precisions_cholesky_ = _compute_precision_cholesky(c, "full")
    np.random.seed(123)
    X_ref, y_ref = gmm.sample(50)
    np.random.seed(123)
    X, y = sample_like_sklearn(50, w, m, c)
    np.testing.assert_array_equal(X, X_ref)
    np.testing.assert_array_equal(y, y_ref)
    with pytest.raises(ValueError):
        sample_like_sklearn(0, w, m, c)


def test_v3_json_maps_to_legacy_like_the_reference_wrapper():
    data = synthetic.make_tiny_primitive()
    v3 = synthetic.to_mgrd_v3_json(data)
    legacy = mgrd_json_to_legacy(v3)
    assert legacy["n_canonical_frames"] == data["n_canonical_frames"]
    np.testing.assert_array_equal(legacy["translation_maxima"], [1, 1, 1])
    assert legacy["eigen_vectors_spatial"] == data["eigen_vectors_spatial"]
    assert "eigen_vectors_time" not in legacy


def _listify(d):
    return {k: (v.tolist() if hasattr(v, "tolist") else v) for k, v in d.items()}


def test_graph_zip_reader(tmp_path):
    """The zip layout and key derivation of ZipReader (reference utilities/zip_io.py:65-233):
    elementary_action_models/elementary_action_<action>/<action>_<primitive>_quaternion_mm.json."""
    prims = [_listify(p) for p in synthetic.make_graph_primitives(3)]
    path = str(tmp_path / "graph.zip")
    actions = {"walk": {"primitives": {"leftStance": prims[0], "rightStance": synthetic.to_mgrd_v3_json(prims[1]),
                                       "idle": {"name": "idle", "spatial_coeffs": [[0.0]], "knots": [0, 0, 0, 0, 1, 1, 1, 1],
                                                "n_canonical_frames": 2}},
                        "info": {"start_states": ["leftStance"], "end_states": ["rightStance"], "idle_states": ["idle"]}},
               "pick": {"primitives": {"reach": prims[2]}, "info": {"start_states": ["reach"], "end_states": ["reach"]}}}
    stored = np.arange(12.0).reshape(3, 4)
    synthetic.write_graph_zip(path, actions, transitions={"walk_leftStance": ["walk_rightStance", "pick_reach"]},
                              start_node=("walk", "walk_leftStance"), cluster_trees={("walk", "leftStance"): stored})
    data = model_io.read_graph_zip(path)
    assert sorted(data["subgraphs"]) == ["pick", "walk"] and data["formatVersion"] == 4.0
    walk = data["subgraphs"]["walk"]
    assert walk["name"] == "walk" and sorted(walk["nodes"]) == ["idle", "leftStance", "rightStance"]
    assert walk["nodes"]["leftStance"]["name"] == "walk_leftStance"            # zip_io.py:186-191
    assert walk["info"]["idle_states"] == ["idle"]
    np.testing.assert_array_equal(walk["nodes"]["leftStance"]["space_partition_json"]["data"], stored)
    assert "space_partition_json" not in walk["nodes"]["rightStance"]
    out = model_io.load_graph_zip(path)
    assert sorted(out) == [("pick", "reach"), ("walk", "leftStance"), ("walk", "rightStance")]   # static idle skipped
    assert out[("walk", "rightStance")]["n_basis_spatial"] == prims[1]["n_basis_spatial"]          # v3 -> legacy
    assert out[("pick", "reach")]["n_canonical_frames"] == prims[2]["n_canonical_frames"]


def test_constraint_conversion_accepts_reference_shaped_objects():
    class Pos(object):
        canonical_keyframe, weight_factor, position, orientation, joint_name = 155, 2.0, [1.0, None, 3.0], None, "Hips"

    class Dir(object):
        canonical_keyframe, weight_factor = 77.5, 1.0
        target_dir = np.array([0.0, 1.0])

    class Pose(object):
        canonical_keyframe, weight_factor = 1, 1.0

    out = constraints_to_device_form([Pos(), Dir(), {"type": "position", "t": 0.0, "weight": 1.0, "target": [0, 0, 0]}], "Hips")
    assert out[0] == {"type": "position", "t": 155.0, "weight": 2.0, "target": [1.0, None, 3.0], "group": 0}
    assert out[1]["type"] == "direction" and out[1]["ref_dir"] == (0.0, 0.0, 1.0) and out[1]["group"] == 1
    with pytest.raises(NotImplementedError):
        constraints_to_device_form([Pose()])

    # one reference constraint -> several device constraints, grouped by the residual entry they add up to:
    # GlobalTransformConstraint = position + orientation in ONE residual (global_transform_constraint.py:70-77),
    # TwoHandConstraint = three residuals (two_hand_constraint.py:66-74)
    class RefSkeleton(object):
        root = "Hips"

    class HandPose(object):
        canonical_keyframe, weight_factor, joint_name, skeleton = 10, 1.0, "LeftHand", RefSkeleton()
        position, orientation = [1.0, 2.0, 3.0], [1.0, 0.0, 0.0, 0.0]

    class TwoHand(object):
        canonical_keyframe, weight_factor = 20, 0.5
        positions, orientations, joint_names = [[0.0, 0.0, 0.0], [2.0, 4.0, 6.0]], [None, None], ["LeftHand", "RightHand"]

    class Relative(object):     # RelativeTransformConstraint: a point in the joint's frame (homogeneous offset)
        canonical_keyframe, weight_factor, joint_name, skeleton = 30, 1.0, "RightHand", RefSkeleton()
        position, orientation, offset = [4.0, 5.0, 6.0], None, [0.0, -3.0, 12.0, 1.0]

    class LookAt(object):
        canonical_keyframe, weight_factor, joint_name = 40, 1.0, "Head"
        target_position = np.array([1.0, 2.0, 3.0])

    class Feet(object):         # FeetConstraint: weight applied inside AND by the caller, one residual entry
        canonical_keyframe, weight_factor = 11, 2.0
        left, right = np.array([1.0, 0.0, 2.0]), np.array([-1.0, 0.0, 2.0])

    out = constraints_to_device_form([Feet()])
    assert [(c["joint"], c["weight"], c["group"], c["target"]) for c in out] == [("LeftFoot", 4.0, 0, [1.0, 0.0, 2.0]), ("RightFoot", 4.0, 0, [-1.0, 0.0, 2.0])]

    class Pose2(object):        # PoseConstraint: cloud, joints, weights, optional velocity of the first joint
        canonical_keyframe, weight_factor = 0, 0.5
        pose_constraint, node_names, weights, velocity_constraint = [[0.0, 1.0, 2.0], [3.0, 4.0, 5.0]], ["Hips", "Head"], [1.0, 2.0], None

    out = constraints_to_device_form([Pose2()])
    assert out == [{"type": "pose", "t": 0.0, "weight": 0.5, "joints": ["Hips", "Head"], "points": [[0.0, 1.0, 2.0], [3.0, 4.0, 5.0]],
                    "weights": [1.0, 2.0], "velocity": None, "group": 0}]

    out = constraints_to_device_form([Relative(), LookAt()])
    assert out[0] == {"type": "joint_position", "t": 30.0, "weight": 1.0, "target": [4.0, 5.0, 6.0], "joint": "RightHand",
                      "offset": [0.0, -3.0, 12.0], "group": 0}
    assert out[1] == {"type": "look_at", "t": 40.0, "weight": 1.0, "target": [1.0, 2.0, 3.0], "joint": "Head", "group": 1}

    out = constraints_to_device_form([HandPose(), TwoHand(), Dir()])
    assert [c["type"] for c in out] == ["joint_position", "joint_orientation", "joint_midpoint", "joint_position", "joint_position", "direction"]
    assert [c["group"] for c in out] == [0, 0, 1, 2, 3, 4]
    assert out[2]["target"] == [1.0, 2.0, 3.0] and out[2]["joint"] == "LeftHand" and out[2]["joint2"] == "RightHand"
    from morphablegraphs_amd.candidate_scoring import group_residuals
    res = np.arange(12.0).reshape(2, 6)
    np.testing.assert_array_equal(group_residuals(out, res), [[1.0, 2.0, 3.0, 4.0, 5.0], [13.0, 8.0, 9.0, 10.0, 11.0]])
    plain = [{"type": "position", "t": 0.0, "weight": 1.0, "target": [0, 0, 0]}] * 6
    assert group_residuals(plain, res) is res


def test_synthetic_models_follow_the_reference_layout():
    d = synthetic.make_walk_primitive(seed=0)
    assert (len(d["eigen_vectors_spatial"]), len(d["eigen_vectors_spatial"][0])) == (40, 31 * 79)
    k = np.array(d["b_spline_knots_spatial"])
    assert len(k) == 35 and np.all(k[:4] == 0) and np.all(k[-4:] == 155)
    assert len(synthetic.make_graph_primitives(16)) == 16


def test_batched_jacobian_reproduces_minpack_forward_differences():
    """HipLeastSquares hands scipy's leastsq a Dfun that evaluates MINPACK lmdif's own forward differences
    (step sqrt(eps) * |x_j|) on one (L + 1)-row batch: on a plain NumPy objective it must land where leastsq's
    built-in differencing lands (reference optimization/least_squares.py:47-50 calls it without Dfun)."""
    from scipy.optimize import leastsq
    from morphablegraphs_amd.motion_primitive_generator import HipLeastSquares
    A = np.random.default_rng(0).standard_normal((7, 4))
    target = np.array([0.3, -1.2, 0.7, 2.0, -0.4, 0.1, 0.9])

    def objective(s, data):                                     # accepts (L,) or (n, L) like the batched objectives
        s = np.asarray(s)
        r = np.tanh(s @ A.T) - data
        return r
    opt = HipLeastSquares({"max_iterations": 500}, objective)
    opt.set_objective_function_parameters(target)
    x0 = np.array([0.1, 0.0, -0.2, 0.3])                         # contains a zero: the absolute-step branch
    got = opt.run(x0)
    ref = leastsq(lambda s: objective(s, target), x0, maxfev=500)[0]
    # same Levenberg-Marquardt iteration up to the stopping test (lmder vs lmdif count evaluations differently)
    np.testing.assert_allclose(got, ref, rtol=0, atol=1e-5)
    f = lambda x: float(np.sum(objective(x, target) ** 2))
    assert abs(f(got) - f(ref)) <= 1e-10 * max(1.0, f(ref))
    J = opt._jac(x0, target)
    h = np.sqrt(np.finfo(float).eps) * np.where(x0 == 0.0, 1.0, np.abs(x0))
    for j in range(4):
        e = np.zeros(4)
        e[j] = h[j]
        np.testing.assert_allclose(J[:, j], (objective(x0 + e, target) - objective(x0, target)) / h[j], rtol=0, atol=1e-6)
    assert J.shape == (7, 4)


def test_per_frame_constraints_of_the_reference_become_frame_device_forms():
    """TrajectoryConstraint on another joint, TrajectorySet / Discrete / Local trajectory, GlobalTransformCA and JointRotation
    constraints (objects shaped like the reference classes) -> the frame_* device forms of frame_constraints.py."""
    from morphablegraphs_amd.frame_constraints import is_frame_constraint, split_frame_constraints
    cps = np.array([[0.0, 0.0, 0.0], [10.0, 0.0, 2.0], [20.0, 1.0, 3.0], [30.0, 1.0, 9.0]])

    class Spline(object):
        control_points = [cps[0]] + list(cps) + [cps[-1], cps[-1]]                    # padded like catmull_rom_spline.py:66-71
        _catmullrom_basematrix = None

    class Skel(object):
        root = "Hips"
        node_name_frame_map = {"Hips": 0, "Spine": 1, "LeftHand": 2}

    class Traj(object):
        constraint_type, joint_name, spline, granularity, weight_factor = "trajectory", "LeftHand", Spline(), 1000, 2.0
        min_arc_length, full_arc_length, skeleton, is_collision_avoidance_constraint = 5.0, 50.0, Skel(), True
        range_start, range_end = 1.0, 9.0

    class RootTraj(Traj):
        joint_name = "Hips"

    class Set(object):
        joint_trajectories, joint_names, joint_arc_lengths, n_canonical_frames, weight_factor = [Traj(), RootTraj()], ["LeftHand", "Hips"], [1.0, 2.0], 47, 1.0

    class Discrete(object):
        joint_name, point_list, unconstrained_indices, weight_factor = "LeftHand", [np.zeros(3), np.ones(3)], None, 1.0

    class Param(object):
        spline, granularity = Spline(), 500

    class Local(object):
        constraint_type, trajectory, start_t, n_canonical_frames, joint_name, weight_factor = "local_trajectory", Param(), 2.5, 47, "Hips", 1.0

    class CA(object):
        constraint_type, joint_name, position, n_canonical_frames, weight_factor, canonical_keyframe = "ca_constraint", "LeftHand", [1.0, None, 2.0], 47, 1.0, 3

    class Rot(object):
        joint_name, rotation_type, rotation_constraint, frame_idx, skeleton, weight_factor = "LeftHand", "euler", [30.0, -20.0, 45.0], 12, Skel(), 1.5

    out = constraints_to_device_form([Traj(), RootTraj(), Set(), Discrete(), Local(), CA(), Rot()])
    assert [c["type"] for c in out] == ["frame_joint_trajectory", "trajectory", "frame_trajectory_set", "frame_discrete_trajectory",
                                        "frame_local_trajectory", "frame_ca_position", "frame_joint_rotation"]
    assert [c["group"] for c in out] == list(range(7))
    fused, frames = split_frame_constraints(out)
    assert len(fused) == 1 and len(frames) == 6 and all(is_frame_constraint(c) for c in frames)
    np.testing.assert_array_equal(out[0]["control_points"], cps)
    assert out[0]["min_u"] == 0.1 and out[0]["joint"] == "LeftHand" and out[0]["weight"] == 2.0
    assert out[2]["trajectories"][0]["range_start"] == 1.0 and out[2]["arc_lengths"] == [1.0, 2.0] and out[2]["n_frames"] == 47
    assert out[3]["unconstrained"] == [0, 1, 2]              # target[None] = 0 zeroes every axis in the reference
    assert out[4]["granularity"] == 500 and out[4]["start_t"] == 2.5
    assert out[5]["target"] == [1.0, None, 2.0]
    assert out[6]["joint_index"] == 2 and out[6]["frame_idx"] == 12.0
    # the Euler target: rotations about the joint's own x, then y, then z (transformations.euler_matrix 'rxyz') = Rx Ry Rz
    from oracle import mg_oracle as orc
    ax, ay, az = np.deg2rad([30.0, -20.0, 45.0])
    Rx = np.array([[1, 0, 0], [0, np.cos(ax), -np.sin(ax)], [0, np.sin(ax), np.cos(ax)]])
    Ry = np.array([[np.cos(ay), 0, np.sin(ay)], [0, 1, 0], [-np.sin(ay), 0, np.cos(ay)]])
    Rz = np.array([[np.cos(az), -np.sin(az), 0], [np.sin(az), np.cos(az), 0], [0, 0, 1]])
    np.testing.assert_allclose(orc.quaternion_matrix3(out[6]["quaternion"]), Rx @ Ry @ Rz, atol=1e-14)


def test_skeleton_description_for_the_c_abi():
    joints, animated = synthetic.make_skeleton()
    sk = _capi.Skeleton(joints, animated)
    assert sk.quat_channel[sk.index("Hips")] == 3 and sk.quat_channel[sk.index("LeftHand_EndSite")] == -1
    assert [sk.names[j] for j in sk.chain("LeftHand")] == ["Hips", "Spine", "Spine1", "LeftShoulder", "LeftArm", "LeftForeArm", "LeftHand"]
    d = sk.desc()
    assert d.n_joints == len(joints) and d.parents and d.offsets and d.quat_channel
    with pytest.raises(ValueError):
        _capi.Skeleton([("A", "B", (0, 0, 0)), ("B", None, (0, 0, 0))], ["A"])    # child before parent
    from morphablegraphs_amd.candidate_scoring import constraints_to_device_form as conv

    class Hand(object):
        canonical_keyframe, weight_factor, position, orientation, joint_name = 10, 1.0, [1.0, 2.0, 3.0], None, "LeftHand"
    assert conv([Hand()], "Hips")[0] == {"type": "joint_position", "t": 10.0, "weight": 1.0, "target": [1.0, 2.0, 3.0], "joint": "LeftHand", "group": 0}


def test_c_abi_header_is_plain_c(tmp_path):
    """include/mg_hip.h is the drop-in boundary: it must compile as strict C99 (and as C++) without any HIP or
    torch header, so that a cgo / JNI / ctypes binding can be generated from it."""
    import subprocess
    src = tmp_path / "hdr.c"
    src.write_text('#include "mg_hip.h"\nint main(void) { mg_primitive_desc d; mg_keyframe_constraint c; mg_skeleton_desc s; '
                   '(void)d; (void)c; (void)s; return MG_OK; }\n')
    inc = os.path.join(ROOT, "include")
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-I", inc, "-fsyntax-only", str(src)])
    subprocess.check_call(["g++", "-std=c++11", "-Wall", "-Werror", "-I", inc, "-fsyntax-only", "-x", "c++", str(src)])


def test_local_optimizer_spends_the_references_evaluation_budget():
    """HipLeastSquares hands MINPACK an analytic-looking Jacobian (the batched finite differences), so scipy runs lmder,
    whose maxfev counts residual calls only; the reference runs leastsq WITHOUT Dfun (least_squares.py:50-53), where the L
    evaluations of every finite-difference Jacobian count too.  Same max_iterations must mean the same stopping point:
    the run is compared with scipy's Dfun-less leastsq on the same objective -- result and evaluation count."""
    from scipy.optimize import leastsq
    from morphablegraphs_amd.motion_primitive_generator import HipLeastSquares
    rng = np.random.default_rng(5)
    L, m = 7, 11
    A = rng.standard_normal((m, L))
    b = rng.standard_normal(m)

    def objective(s, data):                              # works on one vector and on a batch, like the device objectives
        s = np.asarray(s, dtype=np.float64)
        r = np.tanh(s @ A.T) + 0.1 * (s ** 2) @ np.abs(A.T) - b
        return r

    x0 = 0.3 * rng.standard_normal(L)
    for budget in (5, 9, 20, 33, 100, 2000):
        calls = [0]

        def counted(s):
            calls[0] += 1
            return objective(s, None)
        ref = leastsq(counted, x0.copy(), maxfev=budget, full_output=True)
        opt = HipLeastSquares({"max_iterations": budget, "verbose": False}, objective)
        opt.set_objective_function_parameters(None)
        got = opt.run(x0.copy())
        ref_cost, got_cost = np.sum(objective(ref[0], None) ** 2), np.sum(objective(got, None) ** 2)
        assert opt.n_equivalent_evaluations == ref[2]["nfev"], (budget, opt.n_equivalent_evaluations, ref[2]["nfev"])
        # the same iterate, up to what the rounding of a batched residual evaluation does to a finite-difference Jacobian
        np.testing.assert_allclose(got, ref[0], rtol=1e-3, atol=2e-4, err_msg="budget %d" % budget)
        assert abs(got_cost - ref_cost) <= 1e-3 * max(1.0, ref_cost)
    # far fewer launches than evaluations: every Jacobian is one launch
    assert opt.n_launches < opt.n_equivalent_evaluations


def test_graph_zip_reader_follows_format_version_and_the_references_stats_path(tmp_path):
    """read_graph_zip = ZipReader.get_graph_data (reference utilities/zip_io.py:65-233): the layout comes from
    formatVersion (not from the path depth), and '<stem>.stats' is looked up at 'elementary_action_<a>/<stem>.stats'
    without the elementary_action_models prefix (zip_io.py:196); the prefixed location is only a fallback."""
    from morphablegraphs_amd import model_io, synthetic
    prim = synthetic.make_tiny_primitive(seed=2)
    actions = {"walk": {"primitives": {"leftStance": prim}, "info": {"start_states": ["leftStance"]}}}
    st = {"average_step_length": 7.25, "n_standard_transitions": 2}
    for version, prefixed in ((4.0, False), (4.0, True), (2.0, False), (1.0, False)):
        path = str(tmp_path / ("g_%s_%d.zip" % (version, prefixed)))
        synthetic.write_graph_zip(path, actions, format_version=version, node_stats={("walk", "leftStance"): st},
                                  node_stats_prefixed=prefixed)
        data = model_io.read_graph_zip(path)
        node = data["subgraphs"]["walk"]["nodes"]["leftStance"]
        assert node["name"] == "walk_leftStance" and node["stats"] == st, (version, prefixed)
        assert data["subgraphs"]["walk"]["info"] == {"start_states": ["leftStance"]}
    # a version-4 zip read under the wrong assumption must not pick up top-level files as actions
    import json
    import zipfile
    path = str(tmp_path / "mixed.zip")
    synthetic.write_graph_zip(path, actions, format_version=4.0)
    with zipfile.ZipFile(path, "a") as z:
        z.writestr("elementary_action_run/run_fast_quaternion_mm.json", json.dumps(prim))
    assert sorted(model_io.read_graph_zip(path)["subgraphs"]) == ["walk"]


def test_integration_md_names_every_entry_point_of_the_header():
    """INTEGRATION.md section 1 maps EVERY function include/mg_hip.h declares to the reference lines it replaces (the *_host
    convenience variants go with their device-pointer forms)."""
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    header = open(os.path.join(root, "include", "mg_hip.h")).read()
    doc = open(os.path.join(root, "INTEGRATION.md")).read()
    names = sorted(set(re.findall(r"\b(mg_[a-z0-9_]+)\s*\(", header)))
    assert len(names) > 70
    missing = [n for n in names if n not in doc and not n.endswith("_host")]
    assert missing == [], missing


def test_constraint_fingerprint_is_a_value_copy():
    """ADVICE r3 (motion_state_graph.py): the planner step's "same constraints as last step?" test copies values -- arrays,
    nested lists, key names -- so that a target rewritten in place is a different fingerprint, and anything it does not know
    takes the general route."""
    from morphablegraphs_amd.motion_state_graph import constraint_fingerprint as fp
    from morphablegraphs_amd import candidate_scoring as cs
    a = np.array([1.0, 2.0, 3.0])
    clist = [{"type": "position", "t": 5.0, "target": a, "weight": 1.0, "nested": [[1.0, 2.0], [3.0]]}]
    f1 = fp(clist)
    assert f1 == fp([dict(clist[0], target=a.copy())])          # a fresh array with the same values compares equal
    a[0] = 9.0
    f2 = fp(clist)
    assert f1 != f2
    clist[0]["nested"][1][0] = 4.0
    assert fp(clist) != f2
    assert fp([{"type": "position", "t": 5.0, "goal": 1.0}]) != fp([{"type": "position", "t": 5.0, "target": 1.0}])   # key names count
    assert fp([object()]) is None and fp("x") is None and fp([{"type": "position", "target": object()}]) is None
    # the planner step's own test: a flat value copy + the interpreter's comparison
    from morphablegraphs_amd.motion_state_graph import flat_constraint_copy, _same_constraints
    plain = [{"type": "position", "t": 5.0, "weight": 1.0, "target": [10.0, None, 5.0]}, {"type": "direction", "t": 5.0, "target": [0.3, 1.0]}]
    keep = flat_constraint_copy(plain)
    assert _same_constraints(plain, keep) and keep[0]["target"] is not plain[0]["target"]
    plain[0]["target"][0] = -35.0                                   # rewritten in place: noticed
    assert not _same_constraints(plain, keep)
    plain[0]["target"][0] = 10.0
    assert _same_constraints(plain, keep) and not _same_constraints(plain[:1], keep)
    assert not _same_constraints([dict(plain[0], goal=1.0), plain[1]], keep)      # another key
    assert flat_constraint_copy(clist) is None and flat_constraint_copy([{"target": [[1.0]]}]) is None and flat_constraint_copy("x") is None
    assert not _same_constraints([{"target": np.ones(3)}], [{"target": np.ones(3)}])   # arrays: never "the same", no exception
    nan = [{"target": [float("nan"), 1.0]}]
    assert not _same_constraints(nan, flat_constraint_copy([{"target": [float("nan"), 1.0]}]))
    # the shared constraint-set cache: two-dimensional array targets compare without raising, skeletons by serial number
    v1 = cs._values_key([{"weight": 1.0, "target": np.ones((2, 3)), "ref_dir": None}], None)
    v2 = cs._values_key([{"weight": 1.0, "target": np.ones((2, 3)), "ref_dir": None}], None)
    assert not (v1 != v2) and hash(v1) == hash(v2)
    from morphablegraphs_amd import _capi, synthetic
    joints, animated = synthetic.make_skeleton()
    assert _capi.Skeleton(joints, animated).serial != _capi.Skeleton(joints, animated).serial


def test_commands_travel_without_pickle_and_rendezvous_directories_are_private(tmp_path):
    """what crosses the ranks is a length-prefixed JSON document + raw array bytes (nothing a reader
    executes), and the default rendezvous lives in a directory only this user can enter."""
    import stat
    from morphablegraphs_amd import distributed
    cmd = {"op": "options_step", "options": ["a", "b"], "n_samples": 4096, "seed": 7, "dtype": "float32", "skeleton": False,
           "counts": {"a": np.array([10, 20, 30], dtype=np.int64), "b": np.array([60], dtype=np.int64)},
           "constraints": {"a": [{"type": "position", "t": 5.0, "weight": 1.0, "target": [1.0, None, float("nan")]}], "b": []},
           "alignments": {"a": None, "b": {"joint": 0, "heading": [0.0, 1.0], "position": np.array([1.0, 2.0, 3.0])}},
           "samples": np.random.default_rng(0).standard_normal((5, 3)).astype(np.float32), "widths": {"a": 12, "b": np.int64(40)},
           "by_int": {3: "x"}, "node": ("walk", "leftStance"), "by_tuple": {("a", 1): 2.5}}
    blob = distributed.encode_command(cmd)
    assert b"numpy" not in blob and not blob.startswith(b"\x80")          # no pickle
    back = distributed.decode_command(blob)
    assert back["options"] == ["a", "b"] and back["n_samples"] == 4096 and back["alignments"]["a"] is None and back["by_int"] == {3: "x"} and back["node"] == ("walk", "leftStance") and back["by_tuple"] == {("a", 1): 2.5}
    np.testing.assert_array_equal(back["counts"]["a"], cmd["counts"]["a"])
    assert back["samples"].dtype == np.float32 and np.array_equal(back["samples"], cmd["samples"])
    t = back["constraints"]["a"][0]["target"]
    assert t[0] == 1.0 and t[1] is None and np.isnan(t[2]) and back["widths"] == {"a": 12, "b": 40}
    assert distributed.decode_command(distributed.encode_command({"op": "stop"})) == {"op": "stop"}
    with pytest.raises(TypeError):
        distributed.encode_command({"op": "x", "callable": print})
    with pytest.raises(ValueError):
        distributed.decode_command(blob[:-8])                                  # truncated array bytes
    d = distributed.private_rendezvous_dir("mg_rdv_test_%d" % os.getpid())
    try:
        st = os.lstat(d)
        assert stat.S_ISDIR(st.st_mode) and not (st.st_mode & 0o077) and st.st_uid == os.getuid()
        os.chmod(d, 0o755)
        with pytest.raises(PermissionError):
            distributed.private_rendezvous_dir("mg_rdv_test_%d" % os.getpid())
    finally:
        os.rmdir(d)
    # the package imports no tensor framework (north_star: "no PyTorch")
    import subprocess
    pkg = os.path.join(ROOT, "morphablegraphs_amd")
    hits = subprocess.run(["grep", "-rn", "--include=*.py", "torch", pkg], capture_output=True, text=True).stdout
    assert hits == "", hits


def test_optimizer_wrappers_reproduce_the_references_on_toy_objectives():
    """HipLeastSquares / HipNumericalMinimizer against the REFERENCE's own LeastSquares.run / NumericalMinimizer.run
    (least_squares.py:35-64, numerical_minimizer.py:41-76; run by oracle/gen_golden.py on toy objectives, results in
    tests/golden/optimizer_drivers.npz): the point returned for an evaluation budget that cuts the run short (12, 30) and for one
    that does not (400), and scipy.minimize's routes with the wrappers' batched forward differences in place of scipy's own.
    No GPU: the objectives are NumPy."""
    from conftest import load_golden
    from morphablegraphs_amd.motion_primitive_generator import HipLeastSquares, HipNumericalMinimizer
    sys_path_gen = os.path.join(ROOT, "oracle")
    import importlib.util
    spec = importlib.util.spec_from_file_location("gen_golden_toy", os.path.join(sys_path_gen, "gen_golden.py"))
    src = open(spec.origin).read()
    ns = {"np": np}
    start, end = src.index("def toy_residuals("), src.index("def run_optimizer_drivers_case(")
    exec(compile(src[start:end], spec.origin, "exec"), ns)          # the toy objectives themselves, from the generating script
    toy_residuals, toy_scalar = ns["toy_residuals"], ns["toy_scalar"]
    g = load_golden("optimizer_drivers")
    x0 = g["x0"]
    for budget in (12, 30, 400):
        opt = HipLeastSquares({"max_iterations": budget, "verbose": False}, objective=toy_residuals)
        opt.set_objective_function_parameters(None)
        got = opt.run(x0.copy())
        np.testing.assert_allclose(got, g["leastsq_%d" % budget], rtol=0, atol=2e-7, err_msg="budget %d" % budget)
        # the evaluations MINPACK's lmdif would have counted stay within one iteration of the reference's objective calls
        assert abs(opt.n_equivalent_evaluations - int(g["leastsq_calls_%d" % budget])) <= 6, (budget, opt.n_equivalent_evaluations)
    for method, maxiter, tol in (("BFGS", 6, 1e-5), ("BFGS", 200, 1e-5), ("L-BFGS-B", 50, 1e-5), ("Nelder-Mead", 40, 1e-12)):
        st = {"method": method, "max_iterations": maxiter, "diff_eps": 1e-6, "tolerance": 1e-10, "verbose": False}
        opt = HipNumericalMinimizer(st, objective=toy_scalar)
        opt.set_objective_function_parameters(None)
        got = opt.run(x0.copy())
        np.testing.assert_allclose(got, g["minimize_%s_%d" % (method.replace("-", "_"), maxiter)], rtol=0, atol=tol, err_msg=method)
