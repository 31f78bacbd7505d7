"""ctypes binding of libmg_hip.so (include/mg_hip.h) -- the only way host code reaches
the HIP kernels.  There is no CPU fallback: if the library or a gfx950 device is missing
every compute call raises.

The thin object layer here (Context / Primitive / TimeGrid / ConstraintSet / DeviceBuffer)
only owns handles and converts NumPy arrays; the reference-shaped classes live in
motion_primitive.py / motion_spline.py / gaussian_mixture.py.
"""
import ctypes as C
import itertools
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "csrc", "libmg_hip.so")   # the product library; tools pass another build to load_library(path)

MG_OK = 0
MG_ERR_INVALID_ARGUMENT = -1     # enum mg_status (include/mg_hip.h)
MG_F32, MG_F64 = 0, 1
MG_PATH_AUTO, MG_PATH_MFMA, MG_PATH_DIRECT = 0, 1, 2
MG_ALIGN_START_POSE = -1   # mg_alignment_desc.joint: the start-pose branch of the reference's alignment
MG_OPT_FORCE_VALU_SCORE, MG_OPT_FORCE_VALU_SAMPLE, MG_OPT_RING_SLOTS, MG_OPT_CHUNK_WINDOW, MG_OPT_CHUNK_SAMPLES = 0, 1, 2, 3, 4
MG_OPT_FRAMES_KERNEL = 5   # 0 = by batch size, 1 = tile-major, 2 = chunk-stationary
MG_OPT_PLACED_FAST_PCT = 6   # mg_device_malloc_placed's acceptance ratio in percent (tests)
MG_OPT_OPTIONS_STEP = 7      # mg_options_step: 0 = one launch per step where possible, 1 = a chain of launches per option
MG_OPT_PLAIN_MALLOC = 8      # 1 = mg_device_malloc is one hipMalloc whatever the size (no placed output regions)
MG_OPT_GMM_KERNEL = 9        # mg_gmm_log_prob: 0 = by batch size, 1 = one tile per workgroup, 2 = fragments resident in LDS
MG_OPT_SCORE_KERNEL = 10     # mg_score_constraints: 0 = by batch size, 1 = a wave per 16 candidates, 2 = a wave per 64 candidates
MG_OPT_ROOT_MODE = 11        # root channels of the float32 frames kernels: 0, 1 = float64 pipeline (default), 2 = mean/delta split, 3 = split where the gate allows
MG_OPT_PLACED_HOLD = 12      # n > 0: the placement scan holds at most n candidates at once (tests)
MG_OPT_TRAJECTORY_LANES = 13  # closest-point walks: 1 = one lane per candidate whatever the batch, 8 = eight lanes up to 65536 candidates (default: eight while <= 28672 candidates are in flight, four up to 40960)
MG_OPT_TRAJECTORY_SEARCH = 14  # closest-point search of the trajectory constraints: 0 = the reference's (scipy L-BFGS-B restated for one variable), 1 = the monotone walk of rounds 2-4
MG_OPT_COUNT = 15
MG_CONSTRAINT_POSITION, MG_CONSTRAINT_DIRECTION_2D, MG_CONSTRAINT_JOINT_POSITION = 0, 1, 2
MG_CONSTRAINT_JOINT_MIDPOINT, MG_CONSTRAINT_JOINT_ORIENTATION, MG_CONSTRAINT_LOOK_AT, MG_CONSTRAINT_POSE = 3, 4, 5, 6
MG_CONSTRAINT_VALUE_POSITION, MG_CONSTRAINT_VALUE_HEADING = 7, 8   # values of the aligned motion, not errors (chained graph-walk steps)
PROFILE_SLOTS = {"frames": 0, "gmm_log_prob": 1, "score_constraints": 2, "argmin": 3,
                 "gmm_sample": 4, "spline_evaluate": 5, "step": 6, "options_step": 7, "joint_tracks": 8, "frame_constraints": 9, "trajectory": 10}

# every symbol include/mg_hip.h declares (tests check the built library exports them all)
EXPORTED_SYMBOLS = [
    "mg_version", "mg_last_error", "mg_status_string",
    "mg_context_create", "mg_context_destroy", "mg_context_set_stream", "mg_context_set_reserved_cus", "mg_context_set_option", "mg_context_arena_begin", "mg_context_arena_end", "mg_context_arena_bytes", "mg_context_synchronize",
    "mg_dist_unique_id", "mg_dist_init", "mg_dist_all_gather", "mg_dist_finalize", "mg_dist_preflight", "mg_dist_info",
    "mg_context_device_info", "mg_device_malloc", "mg_device_malloc_chunked", "mg_device_malloc_placed", "mg_device_probe_placement", "mg_device_placement_info", "mg_device_free", "mg_context_trim_outputs", "mg_context_output_bytes", "mg_memcpy_h2d", "mg_memcpy_d2h",
    "mg_memset", "mg_profile_enable", "mg_profile_reset", "mg_profile_get", "mg_profile_get_samples",
    "mg_primitive_create", "mg_primitive_destroy", "mg_primitive_info", "mg_primitive_info2", "mg_primitive_root_mode", "mg_primitive_get_precisions_cholesky",
    "mg_time_function_canonical", "mg_time_function_canonical_host", "mg_time_function_sample", "mg_back_project_frames_at",
    "mg_trajectory_create", "mg_trajectory_destroy", "mg_score_trajectory", "mg_score_trajectories", "mg_score_trajectory_points", "mg_trajectory_closest_points", "mg_joint_positions",
    "mg_time_grid_create", "mg_time_grid_destroy", "mg_primitive_canonical_grid", "mg_time_grid_size",
    "mg_time_grid_get_tables",
    "mg_back_project_frames", "mg_back_project_frames_f64", "mg_back_project_coeffs", "mg_spline_evaluate",
    "mg_gmm_log_prob", "mg_gmm_sample", "mg_constraint_set_create", "mg_constraint_set_destroy",
    "mg_score_constraints", "mg_argmin_first", "mg_argmin_first_dev", "mg_step_frames_and_logp", "mg_step_plan", "mg_step_plan_for",
    "mg_back_project_frames_host", "mg_back_project_frames_f64_host", "mg_back_project_coeffs_host",
    "mg_spline_evaluate_host", "mg_gmm_log_prob_host", "mg_gmm_sample_host", "mg_score_constraints_host",
    "mg_score_constraint_residuals", "mg_objective_error_and_naturalness", "mg_gmm_log_prob_jac", "mg_score_constraint_residuals_host",
    "mg_gmm_log_prob_jac_host", "mg_constraint_set_create_fk", "mg_constraint_set_create_aligned", "mg_constraint_set_create_full", "mg_constraint_set_update", "mg_best_candidate", "mg_best_candidate_host",
    "mg_align_frames", "mg_frame_constraint_width", "mg_score_frame_constraint", "mg_score_frame_constraints", "mg_options_frame_lists", "mg_track_plan_create", "mg_track_plan_destroy", "mg_joint_tracks",
    "mg_score_constraint_residuals_chained", "mg_option_step", "mg_options_step", "mg_options_step_device_counts", "mg_option_step_rows", "mg_options_step_rows", "mg_gmm_sample_rows", "mg_dist_broadcast",
]


class MGError(RuntimeError):
    def __init__(self, status, message):
        RuntimeError.__init__(self, "libmg_hip status %d: %s" % (status, message))
        self.status = status


class PrimitiveDesc(C.Structure):
    _fields_ = [("n_basis", C.c_int32), ("n_dim", C.c_int32), ("n_components", C.c_int32),
                ("n_canonical_frames", C.c_int32), ("n_gmm", C.c_int32), ("eigen_is_transposed", C.c_int32),
                ("eigen_vectors", C.c_void_p), ("mean_vector", C.c_void_p), ("translation_maxima", C.c_void_p),
                ("knots", C.c_void_p), ("gmm_weights", C.c_void_p), ("gmm_means", C.c_void_p),
                ("gmm_covars", C.c_void_p), ("n_gmm_dims", C.c_int32), ("n_time_components", C.c_int32),
                ("n_basis_time", C.c_int32), ("reserved", C.c_int32), ("eigen_vectors_time", C.c_void_p),
                ("mean_time_vector", C.c_void_p), ("knots_time", C.c_void_p)]


class KeyframeConstraint(C.Structure):
    _fields_ = [("type", C.c_int32), ("joint", C.c_int32), ("canonical_keyframe", C.c_double),
                ("weight_factor", C.c_double), ("target", C.c_double * 3), ("ref_dir", C.c_double * 3),
                ("joint2", C.c_int32), ("reserved", C.c_int32)]


class PoseConstraintDesc(C.Structure):   # struct mg_pose_constraint
    _fields_ = [("n_points", C.c_int32), ("has_velocity", C.c_int32), ("joints", C.c_void_p), ("points", C.c_void_p),
                ("weights", C.c_void_p), ("velocity", C.c_double * 3)]


class SkeletonDesc(C.Structure):   # struct mg_skeleton_desc
    _fields_ = [("n_joints", C.c_int32), ("reserved", C.c_int32), ("parents", C.c_void_p), ("offsets", C.c_void_p),
                ("quat_channel", C.c_void_p)]


class AlignmentDesc(C.Structure):   # struct mg_alignment_desc
    _fields_ = [("joint", C.c_int32), ("reserved", C.c_int32), ("position", C.c_double * 3), ("heading", C.c_double * 2),
                ("ref_dir", C.c_double * 3)]


MG_FRAME_CA_POSITION, MG_FRAME_DISCRETE_TRAJECTORY, MG_FRAME_LOCAL_TRAJECTORY, MG_FRAME_TRAJECTORY_SET, MG_FRAME_JOINT_ROTATION = 1, 2, 3, 4, 5
MG_FRAME_JOINT_TRAJECTORY = 6      # a TrajectoryConstraint on any joint as a member of a constraint list (start_arc = its min_u)
MG_FRAME_MAX_JOINTS = 8
MG_TRACK_MAX_REQUESTS = 4
MG_FRAME_LIST_MAX = 4        # per-frame constraints of one option in mg_options_frame_lists


class FrameConstraintDesc(C.Structure):   # struct mg_frame_constraint_desc
    _fields_ = [("type", C.c_int32), ("n_frames", C.c_int32), ("n_points", C.c_int32), ("n_joints", C.c_int32), ("weight", C.c_double),
                ("target", C.c_double * 3), ("axis_on", C.c_int32 * 3), ("quat_channel", C.c_int32), ("points_dev", C.c_void_p),
                ("start_arc", C.c_double), ("trajectories", C.c_void_p * MG_FRAME_MAX_JOINTS), ("arc0", C.c_double * MG_FRAME_MAX_JOINTS),
                ("range_start", C.c_double * MG_FRAME_MAX_JOINTS), ("range_end", C.c_double * MG_FRAME_MAX_JOINTS),
                ("has_range", C.c_int32 * MG_FRAME_MAX_JOINTS), ("quaternion", C.c_double * 4)]


def _quat_mul(a, b):
    return np.array([a[0] * b[0] - a[1] * b[1] - a[2] * b[2] - a[3] * b[3], a[0] * b[1] + a[1] * b[0] + a[2] * b[3] - a[3] * b[2],
                     a[0] * b[2] - a[1] * b[3] + a[2] * b[0] + a[3] * b[1], a[0] * b[3] + a[1] * b[2] - a[2] * b[1] + a[3] * b[0]])


def rotate_by_quaternion(q, v):
    """v rotated by the (w, x, y, z) quaternion q (normalised first)."""
    q = np.asarray(q, dtype=np.float64)
    q = q / np.linalg.norm(q)
    v = np.asarray(v, dtype=np.float64)
    u = q[1:]
    return v + 2.0 * (q[0] * np.cross(u, v) + np.cross(u, np.cross(u, v)))


class Skeleton(object):
    """The part of an anim_utils Skeleton forward kinematics needs: joints = [(name, parent name or None,
    (ox, oy, oz)), ...] with parents before children and the root first; animated_joints = names in pose-vector
    order (root translation at [0:3], then one (w,x,y,z) quaternion per animated joint)."""

    _serials = itertools.count(1)

    def __init__(self, joints, animated_joints):
        self.serial = next(Skeleton._serials)   # a stable identity for caches (id() of a collected object can be reused)
        self.names = [j[0] for j in joints]
        index = {n: i for i, n in enumerate(self.names)}
        if len(index) != len(self.names):
            raise ValueError("duplicate joint names")
        self.parents = np.array([-1 if j[1] is None else index[j[1]] for j in joints], dtype=np.int32)
        self.offsets = np.ascontiguousarray([j[2] for j in joints], dtype=np.float64).reshape(len(joints), 3)
        self.animated_joints = list(animated_joints)
        chan = {n: 3 + 4 * i for i, n in enumerate(self.animated_joints)}
        self.quat_channel = np.array([chan.get(n, -1) for n in self.names], dtype=np.int32)
        if self.parents[0] != -1 or np.any(self.parents[1:] >= np.arange(1, len(joints))) or np.any(self.parents[1:] < 0):
            raise ValueError("joint 0 must be the root and parents must precede their children")

    def index(self, joint):
        return int(joint) if isinstance(joint, (int, np.integer)) else self.names.index(joint)

    def chain(self, joint):
        out, j = [], self.index(joint)
        while j >= 0:
            out.insert(0, j)
            j = int(self.parents[j])
        return out

    def heading(self, frame, joint=0, ref_dir=(0.0, 0.0, 1.0)):
        """Unit (x, z) of the joint's global orientation in `frame` applied to ref_dir: what anim_utils'
        get_global_node_orientation_vector returns for one pose vector (host side, once per step: the previous
        motion's last frame; the candidates' own headings are computed on the device)."""
        frame = np.asarray(frame, dtype=np.float64)
        q = np.array([1.0, 0.0, 0.0, 0.0])
        for j in self.chain(joint):
            ch = int(self.quat_channel[j])
            if ch >= 0:
                qj = frame[ch:ch + 4]
                nq = np.linalg.norm(qj)
                if not (np.isfinite(nq) and nq > 0.0):
                    raise ValueError("quaternion of joint %r in the previous frame is zero or not finite" % (self.names[j],))
                q = _quat_mul(q, qj / nq)
        v = np.asarray(ref_dir, dtype=np.float64)
        u = q[1:]
        p = v + 2.0 * (q[0] * np.cross(u, v) + np.cross(u, np.cross(u, v)))
        d = np.array([p[0], p[2]])
        return d / np.linalg.norm(d)

    def alignment_to(self, prev_frame, joint=0, ref_dir=(0.0, 0.0, 1.0)):
        """The alignment record (ConstraintSet `alignment`) that attaches a candidate to a previous motion ending
        in `prev_frame` (= prev_frames[-1] of motion_primitive_constraints.py:110-114)."""
        prev_frame = np.asarray(prev_frame, dtype=np.float64)
        return {"joint": self.index(joint), "position": [float(v) for v in prev_frame[:3]],
                "heading": [float(v) for v in self.heading(prev_frame, joint, ref_dir)], "ref_dir": [float(v) for v in ref_dir]}

    def desc(self):
        return SkeletonDesc(len(self.names), 0, self.parents.ctypes.data, self.offsets.ctypes.data, self.quat_channel.ctypes.data)


_lib = None


def load_library(path=None):
    """dlopen libmg_hip.so; raises OSError with a build hint if it is missing."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    p = path or LIB_PATH
    if not os.path.exists(p):
        raise OSError("libmg_hip.so not found at %s -- build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                      "or `make -C morphablegraphs_amd/csrc` (there is no CPU fallback)" % p)
    # Kernel arguments in device memory instead of host memory the GPU reads over PCIe: the persistent kernels fetch their
    # argument block once per workgroup before anything else can start (0.3-0.5 us of a ~78 us frames launch, tools/ab.py).
    # A setting of the HIP runtime, read when it initialises: it takes effect when this library is the first user of HIP in the
    # process; a value the caller has set is left alone.
    os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")
    lib = C.CDLL(p)
    lib.mg_version.restype = C.c_char_p
    lib.mg_last_error.restype = C.c_char_p
    lib.mg_status_string.restype = C.c_char_p
    lib.mg_status_string.argtypes = [C.c_int]
    lib.mg_primitive_canonical_grid.restype = C.c_void_p
    lib.mg_primitive_canonical_grid.argtypes = [C.c_void_p]
    lib.mg_time_grid_size.argtypes = [C.c_void_p]
    for name in ("mg_context_destroy", "mg_primitive_destroy", "mg_time_grid_destroy", "mg_constraint_set_destroy", "mg_trajectory_destroy", "mg_track_plan_destroy"):
        getattr(lib, name).restype = None
        getattr(lib, name).argtypes = [C.c_void_p]
    vp, i32, i64, u64, dbl = C.c_void_p, C.c_int, C.c_int64, C.c_uint64, C.c_double
    sigs = {
        "mg_context_create": [i32, vp, C.POINTER(vp)],
        "mg_context_set_stream": [vp, vp],
        "mg_context_set_reserved_cus": [vp, i32],
        "mg_context_set_option": [vp, i32, i32],
        "mg_device_malloc_placed": [vp, i64, i32, C.POINTER(vp), C.POINTER(dbl)],
        "mg_device_probe_placement": [vp, vp, i64, C.POINTER(dbl)],
        "mg_device_placement_info": [vp, vp, C.POINTER(dbl), C.POINTER(dbl)],
        "mg_device_malloc_chunked": [vp, i64, i64, C.POINTER(vp)],
        "mg_context_arena_begin": [vp, i64],
        "mg_context_arena_end": [vp],
        "mg_context_arena_bytes": [vp, C.POINTER(i64), C.POINTER(i64)],
        "mg_dist_unique_id": [vp],
        "mg_dist_init": [vp, i32, i32, vp],
        "mg_dist_all_gather": [vp, vp, vp, i64, i32],
        "mg_dist_finalize": [vp],
        "mg_dist_preflight": [vp],
        "mg_dist_info": [vp, C.POINTER(C.c_int32)],
        "mg_context_synchronize": [vp],
        "mg_context_device_info": [vp, C.c_char_p, C.POINTER(C.c_int32), C.POINTER(i64)],
        "mg_device_malloc": [vp, i64, C.POINTER(vp)],
        "mg_context_trim_outputs": [vp],
        "mg_context_output_bytes": [vp, C.POINTER(i64), C.POINTER(i64), C.POINTER(i32), C.POINTER(i32)],
        "mg_device_free": [vp, vp],
        "mg_memcpy_h2d": [vp, vp, vp, i64],
        "mg_memcpy_d2h": [vp, vp, vp, i64],
        "mg_memset": [vp, vp, i32, i64],
        "mg_profile_enable": [vp, i32],
        "mg_profile_reset": [vp],
        "mg_profile_get": [vp, i32, C.POINTER(dbl), C.POINTER(i64)],
        "mg_profile_get_samples": [vp, i32, vp, i64, C.POINTER(i64)],
        "mg_primitive_create": [vp, C.POINTER(PrimitiveDesc), C.POINTER(vp)],
        "mg_primitive_info": [vp, C.POINTER(C.c_int32)],
        "mg_primitive_info2": [vp, C.POINTER(C.c_int32)],
        "mg_primitive_root_mode": [vp, C.POINTER(C.c_int32), C.POINTER(C.c_double)],
        "mg_time_function_canonical": [vp, vp, i32, i64, i64, vp],
        "mg_trajectory_create": [vp, vp, i32, i32, C.POINTER(vp)],
        "mg_joint_positions": [vp, vp, vp, i32, vp, i64, i32, vp],
        "mg_score_trajectory": [vp, vp, vp, vp, i32, i64, i64, dbl, dbl, vp, vp, i32, vp],
        "mg_score_trajectories": [i32, vp, vp, vp, i32, i64, vp, vp, vp, vp, vp, i32],
        "mg_score_trajectory_points": [vp, vp, vp, i64, i32, dbl, dbl, vp, i32, vp],
        "mg_trajectory_closest_points": [vp, vp, vp, i64, i32, dbl, vp, vp, vp],
        "mg_align_frames": [vp, vp, i64, i32, vp, i32, C.POINTER(AlignmentDesc)],
        "mg_frame_constraint_width": [C.POINTER(FrameConstraintDesc), i32],
        "mg_score_frame_constraint": [vp, C.POINTER(FrameConstraintDesc), vp, i64, i32, i32, vp, i32, vp],
        "mg_time_function_canonical_host": [vp, vp, i32, i64, i64, vp],
        "mg_primitive_get_precisions_cholesky": [vp, vp],
        "mg_time_grid_create": [vp, vp, C.c_int32, C.POINTER(vp)],
        "mg_time_grid_get_tables": [vp, vp, vp, vp],
        "mg_back_project_frames": [vp, vp, vp, i32, i64, i64, vp, i32],
        "mg_back_project_frames_f64": [vp, vp, vp, i32, i64, i64, vp],
        "mg_back_project_coeffs": [vp, vp, i32, i64, i64, vp, i32],
        "mg_spline_evaluate": [vp, vp, vp, i64, vp],
        "mg_gmm_log_prob": [vp, vp, i32, i64, i64, vp, i32],
        "mg_gmm_sample": [vp, i64, vp, u64, vp, i32, i64, vp],
        "mg_constraint_set_create": [vp, vp, C.c_int32, C.POINTER(vp)],
        "mg_score_constraints": [vp, vp, vp, i32, i64, i64, vp, i32],
        "mg_argmin_first": [vp, vp, i32, i64, C.POINTER(i64), C.POINTER(dbl)],
        "mg_argmin_first_dev": [vp, vp, i32, i64, vp],
        "mg_step_frames_and_logp": [vp, vp, i32, i64, i64, vp, vp],
        "mg_step_plan": [vp, i64, C.POINTER(C.c_int32)],
        "mg_step_plan_for": [vp, i64, vp, C.POINTER(C.c_int32)],
        "mg_back_project_frames_host": [vp, vp, vp, i32, i64, i64, vp, i32],
        "mg_back_project_frames_f64_host": [vp, vp, vp, i32, i64, i64, vp],
        "mg_back_project_coeffs_host": [vp, vp, i32, i64, i64, vp, i32],
        "mg_spline_evaluate_host": [vp, vp, vp, i64, vp],
        "mg_gmm_log_prob_host": [vp, vp, i32, i64, i64, vp, i32],
        "mg_gmm_sample_host": [vp, i64, vp, u64, vp, i32, i64, vp],
        "mg_score_constraints_host": [vp, vp, vp, i32, i64, i64, vp, i32],
        "mg_score_constraint_residuals": [vp, vp, vp, i32, i64, i64, vp],
        "mg_constraint_set_create_fk": [vp, vp, vp, i32, vp],
        "mg_constraint_set_create_aligned": [vp, vp, vp, i32, vp, vp],
        "mg_constraint_set_update": [vp, vp, i32, vp],
        "mg_constraint_set_create_full": [vp, vp, vp, i32, vp, i32, vp, vp],
        "mg_best_candidate": [vp, vp, vp, i32, i64, i64, C.POINTER(i64), C.POINTER(dbl)],
        "mg_best_candidate_host": [vp, vp, vp, i32, i64, i64, C.POINTER(i64), C.POINTER(dbl)],
        "mg_score_constraint_residuals_chained": [vp, vp, vp, i32, i64, i64, vp, vp],
        "mg_option_step": [vp, vp, i64, vp, u64, vp, i32, i64, vp, vp],
        "mg_option_step_rows": [vp, vp, i64, vp, u64, i64, i64, vp, i32, i64, vp, vp],
        "mg_options_step_rows": [i32, vp, vp, i64, vp, vp, i64, i64, vp, i32, vp, vp, vp, i64, vp],
        "mg_gmm_sample_rows": [vp, i64, vp, u64, i64, i64, vp, i32, i64, vp],
        "mg_dist_broadcast": [vp, vp, i64, i32],
        "mg_objective_error_and_naturalness": [vp, vp, vp, i32, i64, i64, dbl, dbl, vp, vp, vp],
        "mg_time_function_sample": [vp, vp, i32, i64, i64, dbl, vp, vp, i32, vp],
        "mg_back_project_frames_at": [vp, vp, i32, i64, i64, vp, vp, i32, vp, i32],
        "mg_track_plan_create": [vp, vp, i32, vp, vp, i32, C.POINTER(vp)],
        "mg_joint_tracks": [vp, vp, i32, i64, i64, vp, vp, vp],
        "mg_score_frame_constraints": [vp, i32, vp, vp, vp, vp, i64, vp, i32, vp],
        "mg_options_frame_lists": [i32, vp, vp, vp, i32, i64, vp, vp, vp, vp, vp, vp, vp, vp, vp, i64, vp],
        "mg_options_step": [i32, vp, vp, i64, vp, vp, vp, i32, vp, vp, vp, i64, vp],
        "mg_options_step_device_counts": [i32, vp, vp, i64, vp, vp, i32, vp, vp, vp, i64, vp, vp],
        "mg_gmm_log_prob_jac": [vp, vp, i32, i64, i64, vp],
        "mg_score_constraint_residuals_host": [vp, vp, vp, i32, i64, i64, vp],
        "mg_gmm_log_prob_jac_host": [vp, vp, i32, i64, i64, vp],
    }
    for name, argtypes in sigs.items():
        fn = getattr(lib, name)
        fn.argtypes = argtypes
        fn.restype = C.c_int
    if path is None or _lib is None:   # an explicit path given before first use (a tool's diagnostic build) becomes the library
        _lib = lib
    return lib


def _check(status):
    if status != MG_OK:
        raise MGError(status, load_library().mg_last_error().decode("utf-8", "replace"))


def _dtype_code(arr):
    if arr.dtype == np.float32:
        return MG_F32
    if arr.dtype == np.float64:
        return MG_F64
    raise TypeError("expected float32 or float64 array, got %s" % arr.dtype)


def _latents(a):
    """2-D C-contiguous float32/float64 view of a latent batch."""
    a = np.asarray(a)
    if a.dtype not in (np.float32, np.float64):
        a = a.astype(np.float64)
    if a.ndim == 1:
        a = a.reshape(1, -1)
    if a.ndim != 2:
        raise ValueError("latents must be (n_samples, n_components)")
    return np.ascontiguousarray(a)


class Context(object):
    """One per (process, device).  stream: raw hipStream_t (int) or None."""

    def __init__(self, device=0, stream=None, lib=None):
        self.lib = lib if lib is not None else load_library()   # lib: another build loaded with load_library(path) (tools)
        h = C.c_void_p()
        _check(self.lib.mg_context_create(int(device), C.c_void_p(stream) if stream else None, C.byref(h)))
        self.handle = h
        self.device = int(device)

    def close(self):
        if getattr(self, "handle", None):
            self.lib.mg_context_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_reserved_cus(self, n):
        """Leave n CUs free of the persistent frames kernel (for RCCL kernels running beside it)."""
        _check(self.lib.mg_context_set_reserved_cus(self.handle, int(n)))

    def set_option(self, option, value):
        """Test / tuning switches (MG_OPT_*): explicit calls, the library reads no environment variable."""
        _check(self.lib.mg_context_set_option(self.handle, int(option), int(value)))

    def arena_begin(self, block_bytes=0):
        """Until arena_end(), device constants of new primitives come out of shared blocks (one graph, one arena)."""
        _check(self.lib.mg_context_arena_begin(self.handle, int(block_bytes)))

    def arena_end(self):
        _check(self.lib.mg_context_arena_end(self.handle))

    def arena_bytes(self):
        r, u = C.c_int64(0), C.c_int64(0)
        _check(self.lib.mg_context_arena_bytes(self.handle, C.byref(r), C.byref(u)))
        return int(r.value), int(u.value)

    def synchronize(self):
        _check(self.lib.mg_context_synchronize(self.handle))

    # ---- multi-GPU (RCCL loaded on first use; one process per GPU) ----------------------------------
    def dist_unique_id(self):
        """128 opaque bytes made on rank 0 and carried to the other ranks out of band."""
        buf = C.create_string_buffer(128)
        _check(self.lib.mg_dist_unique_id(buf))
        return buf.raw

    def dist_init(self, rank, n_ranks, unique_id):
        _check(self.lib.mg_dist_init(self.handle, int(rank), int(n_ranks), C.c_char_p(bytes(unique_id))))

    def dist_all_gather(self, local_dev, gathered_dev, count, dtype=np.float32):
        """gathered[r * count + i] = rank r's local[i], on the context's stream."""
        code = MG_F64 if np.dtype(dtype) == np.float64 else MG_F32
        _check(self.lib.mg_dist_all_gather(self.handle, _dev_ptr(local_dev), _dev_ptr(gathered_dev), int(count), code))

    def dist_broadcast(self, buf_dev, nbytes, root=0):
        """nbytes bytes of buf_dev from rank `root` to every rank, on the context's stream."""
        _check(self.lib.mg_dist_broadcast(self.handle, _dev_ptr(buf_dev), int(nbytes), int(root)))

    def upload_into(self, buf, arr):
        arr = np.ascontiguousarray(arr)
        _check(self.lib.mg_memcpy_h2d(self.handle, _dev_ptr(buf), arr.ctypes.data_as(C.c_void_p), arr.nbytes))

    def dist_finalize(self):
        _check(self.lib.mg_dist_finalize(self.handle))

    def dist_preflight(self):
        """librccl loads and the context's device answers (mg_dist_preflight): exchanged between the ranks before mg_dist_init."""
        _check(self.lib.mg_dist_preflight(self.handle))

    def dist_info(self):
        """{'rank', 'ranks', 'device'} of the communicator as RCCL reports them (mg_dist_info); ranks 0 without one."""
        out = (C.c_int32 * 3)()
        _check(self.lib.mg_dist_info(self.handle, out))
        return {"rank": int(out[0]), "ranks": int(out[1]), "device": int(out[2])}

    def set_stream(self, stream):
        _check(self.lib.mg_context_set_stream(self.handle, C.c_void_p(stream) if stream else None))

    def device_info(self):
        name = C.create_string_buffer(256)
        ncu, mem = C.c_int32(), C.c_int64()
        _check(self.lib.mg_context_device_info(self.handle, name, C.byref(ncu), C.byref(mem)))
        return {"name": name.value.decode(), "n_cu": ncu.value, "total_mem": mem.value}

    def malloc(self, nbytes, chunk_bytes=0):
        return DeviceBuffer(self, nbytes, chunk_bytes)

    def malloc_placed(self, nbytes, max_candidates=0):
        """A buffer for a large kernel output in the part of the card's memory where the frames kernel's store stream
        runs at the fill rate (mg_device_malloc_placed).  .placement = {"probed", "ratio", "pattern_us", "fast"}."""
        return DeviceBuffer(self, nbytes, placed=True, max_candidates=max_candidates)

    def trim_outputs(self):
        """Release the placed output regions no piece of which is in use (mg_context_trim_outputs)."""
        _check(self.lib.mg_context_trim_outputs(self.handle))

    def output_bytes(self):
        """(reserved, in_use, regions, fast regions) of the context's placed output regions."""
        r, u, n, f = C.c_int64(), C.c_int64(), C.c_int32(), C.c_int32()
        _check(self.lib.mg_context_output_bytes(self.handle, C.byref(r), C.byref(u), C.byref(n), C.byref(f)))
        return r.value, u.value, n.value, f.value

    def probe_placement(self, buf, nbytes=None):
        info = (C.c_double * 4)()
        _check(self.lib.mg_device_probe_placement(self.handle, _dev_ptr(buf), int(nbytes if nbytes is not None else buf.nbytes), info))
        return {"probed": int(info[0]), "ratio": float(info[1]), "pattern_us": float(info[2]), "fast": bool(info[3])}

    def placement_info(self, buf):
        """What the output arena knows about memory it handed out (no probe): 'region' False for memory from elsewhere."""
        info, tbps = (C.c_double * 4)(), C.c_double()
        _check(self.lib.mg_device_placement_info(self.handle, _dev_ptr(buf), info, C.byref(tbps)))
        return {"region": bool(info[0]), "ratio": float(info[1]), "pattern_us": float(info[2]), "fast": bool(info[3]), "pattern_TBps": float(tbps.value)}

    def upload(self, arr):
        arr = np.ascontiguousarray(arr)
        buf = DeviceBuffer(self, arr.nbytes)
        _check(self.lib.mg_memcpy_h2d(self.handle, buf.ptr, arr.ctypes.data_as(C.c_void_p), arr.nbytes))
        return buf

    def download(self, buf, shape, dtype):
        out = np.empty(shape, dtype=dtype)
        ptr = buf.ptr if isinstance(buf, DeviceBuffer) else C.c_void_p(int(buf))
        _check(self.lib.mg_memcpy_d2h(self.handle, out.ctypes.data_as(C.c_void_p), ptr, out.nbytes))
        return out

    def profile_enable(self, on=True):
        """on: False/0 = off, True/1 = bracket every launch, n > 1 = bracket every n-th launch of a slot."""
        _check(self.lib.mg_profile_enable(self.handle, int(on)))

    def profile_reset(self):
        _check(self.lib.mg_profile_reset(self.handle))

    def profile_get(self, slot):
        ms, n = C.c_double(), C.c_int64()
        _check(self.lib.mg_profile_get(self.handle, PROFILE_SLOTS.get(slot, slot), C.byref(ms), C.byref(n)))
        return ms.value, n.value

    def profile_samples(self, slot, capacity=65536):
        """The individual event-bracketed durations (ms) of a slot, oldest first."""
        out = np.empty(int(capacity), dtype=np.float32)
        n = C.c_int64()
        _check(self.lib.mg_profile_get_samples(self.handle, PROFILE_SLOTS.get(slot, slot), out.ctypes.data_as(C.c_void_p),
                                               int(capacity), C.byref(n)))
        return out[:n.value].copy()

    def joint_positions(self, skeleton, joints, frames):
        """(N, D) float64 frames -> (N, len(joints), 3) global joint positions by forward kinematics (mg_joint_positions);
        joints: names or indices of `skeleton` (a Skeleton)."""
        F = np.ascontiguousarray(frames, dtype=np.float64)
        if F.ndim != 2:
            raise ValueError("frames must be (n_frames, n_dim)")
        idx = np.ascontiguousarray([skeleton.index(j) for j in joints], dtype=np.int32)
        out = np.empty((F.shape[0], len(idx), 3), dtype=np.float64)
        d_f, d_o = self.upload(F), self.malloc(max(out.nbytes, 8))
        try:
            d = skeleton.desc()
            _check(self.lib.mg_joint_positions(self.handle, C.byref(d), idx.ctypes.data_as(C.c_void_p), len(idx), d_f.ptr, F.shape[0], F.shape[1], d_o.ptr))
            return self.download(d_o, out.shape, np.float64)
        finally:
            d_f.free()
            d_o.free()

    def argmin_first(self, values_dev, n, dtype=np.float32):
        idx, val = C.c_int64(), C.c_double()
        ptr = values_dev.ptr if isinstance(values_dev, DeviceBuffer) else C.c_void_p(int(values_dev))
        code = MG_F64 if np.dtype(dtype) == np.float64 else MG_F32
        _check(self.lib.mg_argmin_first(self.handle, ptr, code, int(n), C.byref(idx), C.byref(val)))
        return idx.value, val.value


class DeviceBuffer(object):
    def __init__(self, ctx, nbytes, chunk_bytes=0, placed=False, max_candidates=0):
        """chunk_bytes > 0: assembled from separate physical chunks of that size (mg_device_malloc_chunked);
        placed: probed for the fast placement class (mg_device_malloc_placed)."""
        self.ctx = ctx
        self.nbytes = int(nbytes)
        self.placement = None
        p = C.c_void_p()
        if placed:
            info = (C.c_double * 4)()
            _check(ctx.lib.mg_device_malloc_placed(ctx.handle, self.nbytes, int(max_candidates), C.byref(p), info))
            self.placement = {"probed": int(info[0]), "ratio": float(info[1]), "pattern_us": float(info[2]), "fast": bool(info[3])}
        elif chunk_bytes:
            _check(ctx.lib.mg_device_malloc_chunked(ctx.handle, self.nbytes, int(chunk_bytes), C.byref(p)))
        else:
            _check(ctx.lib.mg_device_malloc(ctx.handle, self.nbytes, C.byref(p)))
        self.ptr = p

    @property
    def address(self):
        return self.ptr.value

    def free(self):
        if getattr(self, "ptr", None) and self.ctx.handle:
            self.ctx.lib.mg_device_free(self.ctx.handle, self.ptr)
        self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


def _dev_ptr(x):
    if isinstance(x, DeviceBuffer):
        return x.ptr
    return C.c_void_p(int(x))


class TimeGrid(object):
    def __init__(self, prim, times=None, handle=None):
        self.prim = prim
        self.owned = handle is None
        if handle is None:
            t = np.ascontiguousarray(np.atleast_1d(times), dtype=np.float64)
            h = C.c_void_p()
            _check(prim.lib.mg_time_grid_create(prim.handle, t.ctypes.data_as(C.c_void_p), len(t), C.byref(h)))
            handle = h
        self.handle = handle
        self.size = prim.lib.mg_time_grid_size(self.handle)

    def tables(self):
        i0 = np.empty(self.size, dtype=np.int32)
        w = np.empty((self.size, 4))
        t = np.empty(self.size)
        _check(self.prim.lib.mg_time_grid_get_tables(self.handle, i0.ctypes.data_as(C.c_void_p),
                                                     w.ctypes.data_as(C.c_void_p), t.ctypes.data_as(C.c_void_p)))
        return i0, w, t

    def close(self):
        if self.owned and getattr(self, "handle", None) and self.prim.handle and self.prim.ctx.handle:
            self.prim.lib.mg_time_grid_destroy(self.handle)
        self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class ConstraintSet(object):
    """constraints: list of dicts {"type": "position"|"direction"|"joint_position"|"joint_midpoint"|
    "joint_orientation"|"look_at"|"pose" ({"joints": names, "points": (N, 3), "weights": (N,), "velocity": xyz or None}), "t": float, "weight": float, "target": [x|None, y|None, z|None] | [dx, dz],
    "ref_dir": (rx, ry, rz), "joint": name or index, "joint2": second joint of a midpoint, "orientation": wanted
    global (w,x,y,z) of a joint_orientation (or "target": that orientation applied to ref_dir), "offset": a point in
    the joint's own frame instead of its origin (joint_position), "look_at": {"joint", "target" xyz}}; a `skeleton`
    (Skeleton) is needed for "joint_position", "joint_midpoint" and for orientations of joints other than the root.  `alignment` = {"joint", "position", "heading",
    "ref_dir"} (Skeleton.alignment_to) switches to global coordinates: every candidate is aligned to the previous
    motion before its constraints are evaluated; without a skeleton the aligning node is the root joint."""

    def __init__(self, prim, constraints, skeleton=None, alignment=None):
        self.prim = prim
        self.skeleton = skeleton
        self.alignment = alignment
        n = len(constraints)
        arr = self._marshal(constraints, skeleton)
        h = C.c_void_p()
        poses = [c for c in constraints if c["type"] == "pose"]
        if poses:
            if skeleton is None:
                raise ValueError("pose constraints need a skeleton")
            keep = []                                    # arrays the descriptors point at
            parr = (PoseConstraintDesc * len(poses))()
            for i, c in enumerate(poses):
                joints = np.ascontiguousarray([skeleton.index(j) for j in c["joints"]], dtype=np.int32)
                pts = np.ascontiguousarray(c["points"], dtype=np.float64).reshape(len(joints), 3)
                wts = np.ascontiguousarray(c.get("weights", np.ones(len(joints))), dtype=np.float64).reshape(len(joints))
                keep += [joints, pts, wts]
                parr[i].n_points, parr[i].has_velocity = len(joints), int(c.get("velocity") is not None)
                parr[i].joints, parr[i].points, parr[i].weights = joints.ctypes.data, pts.ctypes.data, wts.ctypes.data
                for a in range(3):
                    parr[i].velocity[a] = float(c["velocity"][a]) if c.get("velocity") is not None else 0.0
            al = self._marshal_alignment(alignment, skeleton) if alignment is not None else None
            d = skeleton.desc()
            _check(prim.lib.mg_constraint_set_create_full(prim.handle, C.byref(d), C.cast(arr, C.c_void_p), n, C.cast(parr, C.c_void_p),
                                                          len(poses), C.byref(al) if al is not None else None, C.byref(h)))
        elif alignment is not None:
            al = self._marshal_alignment(alignment, skeleton)
            d = skeleton.desc() if skeleton is not None else None
            _check(prim.lib.mg_constraint_set_create_aligned(prim.handle, C.byref(d) if d is not None else None,
                                                             C.cast(arr, C.c_void_p), n, C.byref(al), C.byref(h)))
        elif skeleton is None:
            _check(prim.lib.mg_constraint_set_create(prim.handle, C.cast(arr, C.c_void_p), n, C.byref(h)))
        else:
            d = skeleton.desc()
            _check(prim.lib.mg_constraint_set_create_fk(prim.handle, C.byref(d), C.cast(arr, C.c_void_p), n, C.byref(h)))
        self.handle = h
        self.n = n

    def update(self, constraints, alignment=None):
        """New targets / weights / previous frame for the same STRUCTURE (types, joints, keyframes, relative points,
        aligning joint): one small stream-ordered launch instead of a new set (mg_constraint_set_update)."""
        arr = self._marshal(constraints, self.skeleton)
        al = self._marshal_alignment(alignment, self.skeleton) if alignment is not None else None
        _check(self.prim.lib.mg_constraint_set_update(self.handle, C.cast(arr, C.c_void_p), len(constraints),
                                                      C.byref(al) if al is not None else None))
        self.alignment = alignment

    @staticmethod
    def _marshal_alignment(alignment, skeleton):
        al = AlignmentDesc()
        j = alignment.get("joint", 0)
        al.joint = int(j) if (skeleton is None or j == MG_ALIGN_START_POSE) else skeleton.index(j)
        for a in range(3):
            al.position[a] = float(alignment["position"][a])
            al.ref_dir[a] = float(alignment.get("ref_dir", (0.0, 0.0, 1.0))[a])
        al.heading[0], al.heading[1] = float(alignment["heading"][0]), float(alignment["heading"][1])
        return al

    @staticmethod
    def _marshal(constraints, skeleton):
        n = len(constraints)
        arr = (KeyframeConstraint * max(n, 1))()
        for i, c in enumerate(constraints):
            k = arr[i]
            k.canonical_keyframe = float(c["t"])
            k.weight_factor = float(c.get("weight", 1.0))
            if c["type"] == "position":
                k.type = MG_CONSTRAINT_POSITION
                for a in range(3):
                    v = c["target"][a]
                    k.target[a] = float("nan") if v is None else float(v)
            elif c["type"] == "joint_position":
                if skeleton is None:
                    raise ValueError("joint_position constraints need a skeleton")
                k.type = MG_CONSTRAINT_JOINT_POSITION
                k.joint = skeleton.index(c["joint"])
                for a in range(3):
                    v = c["target"][a]
                    k.target[a] = float("nan") if v is None else float(v)
                    k.ref_dir[a] = float(c["offset"][a]) if c.get("offset") is not None else 0.0   # point in the joint's frame
            elif c["type"] == "look_at":
                if skeleton is None and c.get("joint", 0) not in (0, None):
                    raise ValueError("look_at on joint %r needs a skeleton (only the root joint, 0, does not)" % (c["joint"],))
                k.type = MG_CONSTRAINT_LOOK_AT
                k.joint = 0 if skeleton is None else skeleton.index(c.get("joint", 0) or 0)
                rd = c.get("ref_dir", (0.0, 0.0, 1.0))
                for a in range(3):
                    k.target[a] = float(c["target"][a])
                    k.ref_dir[a] = float(rd[a])
            elif c["type"] == "joint_midpoint":
                if skeleton is None:
                    raise ValueError("joint_midpoint constraints need a skeleton")
                k.type = MG_CONSTRAINT_JOINT_MIDPOINT
                k.joint, k.joint2 = skeleton.index(c["joint"]), skeleton.index(c["joint2"])
                for a in range(3):
                    v = c["target"][a]
                    k.target[a] = float("nan") if v is None else float(v)
            elif c["type"] == "joint_orientation":
                k.type = MG_CONSTRAINT_JOINT_ORIENTATION
                if skeleton is None and c.get("joint", 0) not in (0, None):
                    raise ValueError("the orientation of joint %r needs a skeleton (only the root joint, 0, does not)" % (c["joint"],))
                k.joint = 0 if skeleton is None else skeleton.index(c.get("joint", 0) or 0)
                rd = c.get("ref_dir", (0.0, 0.0, 1.0))
                tv = rotate_by_quaternion(c["orientation"], rd) if "orientation" in c else c["target"]
                for a in range(3):
                    k.ref_dir[a] = float(rd[a])
                    k.target[a] = float(tv[a])
            elif c["type"] == "pose":
                k.type = MG_CONSTRAINT_POSE
                k.joint = sum(1 for q in constraints[:i] if q["type"] == "pose")   # index into the pose array
            elif c["type"] == "value_position":      # {"type", "t", "axis": 0 | 1 | 2}: the (aligned) root position's component
                k.type = MG_CONSTRAINT_VALUE_POSITION
                k.target[0] = float(int(c["axis"]))
            elif c["type"] == "value_heading":       # {"type", "t", "axis": 0 | 2, "joint", "ref_dir"}: the (aligned) unit heading's component
                k.type = MG_CONSTRAINT_VALUE_HEADING
                if skeleton is None and c.get("joint", 0) not in (0, None):
                    raise ValueError("the heading of joint %r needs a skeleton (only the root joint, 0, does not)" % (c["joint"],))
                k.joint = 0 if skeleton is None else skeleton.index(c.get("joint", 0) or 0)
                k.target[0] = float(int(c["axis"]))
                rd = c.get("ref_dir", (0.0, 0.0, 1.0))
                for a in range(3):
                    k.ref_dir[a] = float(rd[a])
            elif c["type"] == "direction":
                k.type = MG_CONSTRAINT_DIRECTION_2D
                k.target[0], k.target[1], k.target[2] = float(c["target"][0]), float(c["target"][1]), 0.0
                rd = c.get("ref_dir", (0.0, 0.0, 1.0))
                for a in range(3):
                    k.ref_dir[a] = float(rd[a])
            else:
                raise ValueError("unknown constraint type %r" % (c["type"],))
        return arr

    def close(self):
        # the C object points at its primitive and context: never touch it after either of them has been destroyed
        if getattr(self, "handle", None) and self.prim.handle and self.prim.ctx.handle:
            self.prim.lib.mg_constraint_set_destroy(self.handle)
        self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Trajectory(object):
    """A Catmull-Rom trajectory on the device for mg_score_trajectory: control_points (n, 3), the reference's
    TrajectoryConstraint spline (trajectory_constraint.py:33-38); granularity = the step of the closest-point search."""

    def __init__(self, prim, control_points, granularity=1000):
        self.prim = prim
        cp = np.ascontiguousarray(np.asarray(control_points, dtype=np.float64))
        if cp.ndim != 2 or cp.shape[1] != 3:
            raise ValueError("control_points must be (n, 3)")
        h = C.c_void_p()
        _check(prim.lib.mg_trajectory_create(prim.handle, cp.ctypes.data_as(C.c_void_p), cp.shape[0], int(granularity), C.byref(h)))
        self.handle = h

    def close(self):
        if getattr(self, "handle", None) and self.prim.handle and self.prim.ctx.handle:
            self.prim.lib.mg_trajectory_destroy(self.handle)
        self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class TrackPlan(object):
    """What mg_joint_tracks needs that does not change between calls (mg_track_plan_create): per request its joints' chains; the
    channels to keep; the aligning node's chain.  requests: [[joint, ...], ...] (indices or names of `skeleton`), at most 4."""

    def __init__(self, prim, skeleton, requests, align_joint=0):
        self.prim, self.skeleton = prim, skeleton
        self.requests = [[skeleton.index(j) for j in r] for r in requests]
        nj = np.ascontiguousarray([len(r) for r in self.requests], dtype=np.int32)
        flat = np.ascontiguousarray([j for r in self.requests for j in r], dtype=np.int32)
        d = skeleton.desc()
        h = C.c_void_p()
        _check(prim.lib.mg_track_plan_create(prim.handle, C.byref(d), len(self.requests), nj.ctypes.data_as(C.c_void_p), flat.ctypes.data_as(C.c_void_p),
                                             int(align_joint), C.byref(h)))
        self.handle = h

    def tracks_dev(self, lat_dev, lat_dtype, n, ld, grids, out_devs, alignment=None):
        """grids: one TimeGrid or None (canonical) per request; out_devs: one device buffer (n, T, joints, 3) float64 per request."""
        m = len(self.requests)
        g = (C.c_void_p * m)(*[(x.handle if x is not None else None) for x in grids])
        o = (C.c_void_p * m)(*[_dev_ptr(x).value for x in out_devs])
        al = ConstraintSet._marshal_alignment(alignment, self.skeleton) if alignment is not None else None
        lc = MG_F64 if np.dtype(lat_dtype) == np.float64 else MG_F32
        _check(self.prim.lib.mg_joint_tracks(self.handle, _dev_ptr(lat_dev), lc, int(n), int(ld), C.byref(al) if al is not None else None, g, o))

    def close(self):
        if getattr(self, "handle", None) and self.prim.handle and self.prim.ctx.handle:
            self.prim.lib.mg_track_plan_destroy(self.handle)
        self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Primitive(object):
    """Device-resident constants of one motion primitive, built from the reference's JSON dict."""
    _serials = itertools.count(1)

    def __init__(self, ctx, data):
        self.serial = next(Primitive._serials)   # a stable identity for caches (id() and the handle's address can both be reused)
        self.ctx = ctx
        self.lib = ctx.lib
        eig = np.ascontiguousarray(np.asarray(data["eigen_vectors_spatial"], dtype=np.float64))
        mean = np.ascontiguousarray(np.asarray(data["mean_spatial_vector"], dtype=np.float64))
        knots = np.ascontiguousarray(np.asarray(data["b_spline_knots_spatial"], dtype=np.float64))
        tm = np.ascontiguousarray(np.asarray(data.get("translation_maxima", [1.0, 1.0, 1.0]), dtype=np.float64))
        nb, nd = int(data["n_basis_spatial"]), int(data["n_dim_spatial"])
        if eig.ndim != 2 or eig.shape[1] != nb * nd:
            raise ValueError("eigen_vectors_spatial must be (n_components, n_basis*n_dim)")
        if mean.shape != (nb * nd,) or knots.shape != (nb + 4,) or tm.shape != (3,):
            raise ValueError("mean/knots/translation_maxima have the wrong shape")
        d = PrimitiveDesc()
        d.n_basis, d.n_dim, d.n_components = nb, nd, eig.shape[0]
        d.n_canonical_frames = int(data["n_canonical_frames"])
        d.eigen_is_transposed = 0
        d.eigen_vectors = eig.ctypes.data
        d.mean_vector = mean.ctypes.data
        d.translation_maxima = tm.ctypes.data
        d.knots = knots.ctypes.data
        keep = [eig, mean, knots, tm]
        if "gmm_weights" in data and data["gmm_weights"] is not None:
            gw = np.ascontiguousarray(np.asarray(data["gmm_weights"], dtype=np.float64))
            gm = np.ascontiguousarray(np.asarray(data["gmm_means"], dtype=np.float64))
            gc = np.ascontiguousarray(np.asarray(data["gmm_covars"], dtype=np.float64))
            # the reference fits the mixture over the concatenated (spatial, time) latents: Lg >= L columns
            L = eig.shape[0]
            Lg = gm.shape[1] if gm.ndim == 2 else -1
            if gm.ndim != 2 or gm.shape[0] != len(gw) or Lg < L or gc.shape != (len(gw), Lg, Lg):
                raise ValueError("gmm_means/gmm_covars have the wrong shape")
            d.n_gmm = len(gw)
            d.n_gmm_dims = Lg
            d.gmm_weights, d.gmm_means, d.gmm_covars = gw.ctypes.data, gm.ctypes.data, gc.ctypes.data
            keep += [gw, gm, gc]
        else:
            d.n_gmm = 0
        if data.get("eigen_vectors_time") is not None:
            et = np.ascontiguousarray(np.asarray(data["eigen_vectors_time"], dtype=np.float64))
            mt = np.ascontiguousarray(np.asarray(data["mean_time_vector"], dtype=np.float64))
            kt = np.ascontiguousarray(np.asarray(data["b_spline_knots_time"], dtype=np.float64))
            nbt = int(data["n_basis_time"])
            if et.ndim != 2 or et.shape[0] != nbt or mt.shape != (nbt,) or kt.shape != (nbt + 4,):
                raise ValueError("eigen_vectors_time must be (n_basis_time, n_time_components); mean_time_vector (n_basis_time); knots (n_basis_time + 4)")
            d.n_time_components, d.n_basis_time = et.shape[1], nbt
            d.eigen_vectors_time, d.mean_time_vector, d.knots_time = et.ctypes.data, mt.ctypes.data, kt.ctypes.data
            keep += [et, mt, kt]
        h = C.c_void_p()
        _check(self.lib.mg_primitive_create(ctx.handle, C.byref(d), C.byref(h)))
        self.handle = h
        info = (C.c_int32 * 8)()
        _check(self.lib.mg_primitive_info(self.handle, info))
        (self.n_basis, self.n_dim, self.n_components, self.n_canonical_frames, self.n_gmm,
         self.kk, mfma, self.n_chunks) = [int(v) for v in info]
        self.mfma_supported = bool(mfma)
        info2 = (C.c_int32 * 4)()
        _check(self.lib.mg_primitive_info2(self.handle, info2))
        self.n_gmm_dims, self.n_time_components, self.n_basis_time, self.kk_gmm = [int(v) for v in info2]
        est = C.c_double()
        _check(self.lib.mg_primitive_root_mode(self.handle, None, C.byref(est)))
        self.root_split_estimate = float(est.value)   # the mean/delta split's error estimate (mg_primitive_root_mode)
        self.canonical_grid = TimeGrid(self, handle=C.c_void_p(self.lib.mg_primitive_canonical_grid(self.handle)))

    @property
    def root_split(self):
        """True when the float32 frames kernels compute this primitive's root channels by the mean/delta split (MG_OPT_ROOT_MODE
        2, or 3 where the accuracy gate of mg_primitive_root_mode allows it), False for the float64 pipeline (the default)."""
        split = C.c_int32()
        _check(self.lib.mg_primitive_root_mode(self.handle, C.byref(split), None))
        return bool(split.value)

    def close(self):
        if getattr(self, "handle", None) and self.ctx.handle:
            self.lib.mg_primitive_destroy(self.handle)
        self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- queries -----------------------------------------------------------------------
    def precisions_cholesky(self):
        out = np.empty((self.n_gmm, self.n_gmm_dims, self.n_gmm_dims))
        _check(self.lib.mg_primitive_get_precisions_cholesky(self.handle, out.ctypes.data_as(C.c_void_p)))
        return out

    def score_trajectory_dev(self, trajectory, lat_dev, lat_dtype, n, ld, errors_dev, min_u=0.0, weight=1.0, alignment=None,
                             accumulate=False, residuals_dev=None, grid=None):
        """mg_score_trajectory on device buffers: errors_dev (n) float64 written or added to."""
        al = ConstraintSet._marshal_alignment(alignment, None) if alignment is not None else None
        code = MG_F64 if np.dtype(lat_dtype) == np.float64 else MG_F32
        _check(self.lib.mg_score_trajectory(self.handle, trajectory.handle, self._grid_handle(grid), _dev_ptr(lat_dev), code, int(n), int(ld),
                                            float(min_u), float(weight), C.byref(al) if al is not None else None, _dev_ptr(errors_dev),
                                            1 if accumulate else 0, _dev_ptr(residuals_dev) if residuals_dev is not None else None))

    @staticmethod
    def score_trajectories_dev(prims, trajectories, lat_devs, lat_dtype, n, lds, err_devs, min_us, weights, alignments=None, accumulate=False):
        """mg_score_trajectories: the k-th primitive's n resident candidates against the k-th trajectory, all in one launch (primitives of
        one context); err_devs[k] (n,) float64 written or added to."""
        m = len(prims)
        vp = C.c_void_p
        keep = [ConstraintSet._marshal_alignment(a, None) if a is not None else None for a in (alignments or [None] * m)]
        al = (vp * m)(*[(C.addressof(a) if a is not None else None) for a in keep]) if alignments is not None else None
        code = MG_F64 if np.dtype(lat_dtype) == np.float64 else MG_F32
        _check(prims[0].lib.mg_score_trajectories(m, (vp * m)(*[p.handle.value for p in prims]), (vp * m)(*[t.handle.value for t in trajectories]),
                                                  (vp * m)(*[_dev_ptr(x).value for x in lat_devs]), code, int(n), (C.c_int64 * m)(*[int(x) for x in lds]),
                                                  (C.c_double * m)(*[float(x) for x in min_us]), (C.c_double * m)(*[float(x) for x in weights]), al,
                                                  (vp * m)(*[_dev_ptr(x).value for x in err_devs]), 1 if accumulate else 0))

    def score_trajectory(self, trajectory, S, min_u=0.0, weight=1.0, alignment=None, residuals=False, grid=None):
        """(n,) float64 errors = weight * average distance of the root path to the trajectory; with residuals=True also
        the (n, T) per-sample residuals (TrajectoryConstraint.get_residual_vector times the weight)."""
        S = _latents(S)
        n, T = S.shape[0], self._grid_size(grid)
        d_S, d_e = self.ctx.upload(S), self.ctx.malloc(max(n, 1) * 8)
        d_r = self.ctx.malloc(max(n * T, 1) * 8) if residuals else None
        try:
            self.score_trajectory_dev(trajectory, d_S, S.dtype, n, S.shape[1], d_e, min_u, weight, alignment, False, d_r, grid)
            err = self.ctx.download(d_e, (n,), np.float64)
            return (err, self.ctx.download(d_r, (n, T), np.float64)) if residuals else err
        finally:
            for b in (d_S, d_e, d_r):
                if b is not None:
                    b.free()

    def joint_tracks(self, skeleton, joints, S, grid=None):
        """(B, T, len(joints), 3) float64: the global positions of `joints` in every frame of every sample's motion -- float64
        back projection (mg_back_project_frames_f64) and forward kinematics (mg_joint_positions) on the device, only the
        tracks come back.  What the per-frame constraints of the reference walk a motion for (trajectory constraints on other
        joints than the root, collision avoidance, local and discrete trajectories)."""
        S = _latents(S)
        n, T = S.shape[0], self._grid_size(grid)
        idx = np.ascontiguousarray([skeleton.index(j) for j in joints], dtype=np.int32)
        ctx = self.ctx
        d_S = ctx.upload(S)
        d_f = ctx.malloc(max(n * T * self.n_dim, 1) * 8)
        d_o = ctx.malloc(max(n * T * len(idx) * 3, 1) * 8)
        try:
            _check(self.lib.mg_back_project_frames_f64(self.handle, self._grid_handle(grid), d_S.ptr, _dtype_code(S), n, S.shape[1], d_f.ptr))
            d = skeleton.desc()
            _check(self.lib.mg_joint_positions(ctx.handle, C.byref(d), idx.ctypes.data_as(C.c_void_p), len(idx), d_f.ptr, n * T, self.n_dim, d_o.ptr))
            return ctx.download(d_o, (n, T, len(idx), 3), np.float64)
        finally:
            for b in (d_S, d_f, d_o):
                b.free()

    def trajectory_closest_points(self, trajectory, points, min_u=0.0, evaluations=False):
        """mg_trajectory_closest_points: points (n, T, 3) float64 -> (parameters (n, T), distances (n, T)): the reference's
        find_closest_point_fast chained frame to frame (trajectory_constraint.py:103-113) for every row; evaluations: also the
        (f, g) evaluations every search took (n, T) int32."""
        P = np.ascontiguousarray(points, dtype=np.float64)
        n, T = P.shape[0], P.shape[1]
        ctx = self.ctx
        d_p, d_u, d_d = ctx.upload(P), ctx.malloc(max(n * T, 1) * 8), ctx.malloc(max(n * T, 1) * 8)
        d_n = ctx.malloc(max(n * T, 1) * 4) if evaluations else None
        try:
            if d_n is not None:
                _check(self.lib.mg_memset(ctx.handle, d_n.ptr, 0, max(n * T, 1) * 4))
            _check(self.lib.mg_trajectory_closest_points(self.handle, trajectory.handle, d_p.ptr, n, T, float(min_u), d_u.ptr, d_d.ptr,
                                                         d_n.ptr if d_n is not None else None))
            out = (ctx.download(d_u, (n, T), np.float64), ctx.download(d_d, (n, T), np.float64))
            return out + (ctx.download(d_n, (n, T), np.int32),) if evaluations else out
        finally:
            for b in (d_p, d_u, d_d, d_n):
                if b is not None:
                    b.free()

    def score_trajectory_points(self, trajectory, points, min_u=0.0, weight=1.0, residuals=False):
        """mg_score_trajectory_points: points (n, T, 3) float64 followed along the trajectory instead of the root path."""
        P = np.ascontiguousarray(points, dtype=np.float64)
        n, T = P.shape[0], P.shape[1]
        ctx = self.ctx
        d_p, d_e = ctx.upload(P), ctx.malloc(max(n, 1) * 8)
        d_r = ctx.malloc(max(n * T, 1) * 8) if residuals else None
        try:
            _check(self.lib.mg_score_trajectory_points(self.handle, trajectory.handle, d_p.ptr, n, T, float(min_u), float(weight), d_e.ptr, 0,
                                                       d_r.ptr if d_r is not None else None))
            err = ctx.download(d_e, (n,), np.float64)
            return (err, ctx.download(d_r, (n, T), np.float64)) if residuals else err
        finally:
            for b in (d_p, d_e, d_r):
                if b is not None:
                    b.free()

    def time_function_canonical(self, gamma):
        """(B, n_time_components) time latents -> (B, n_canonical_frames) float64: the reference's
        _back_transform_gamma_to_canonical_time_function (motion_primitive.py:289-302) for every row."""
        G = _latents(gamma)
        out = np.empty((G.shape[0], self.n_canonical_frames), dtype=np.float64)
        _check(self.lib.mg_time_function_canonical_host(self.handle, G.ctypes.data_as(C.c_void_p), _dtype_code(G), G.shape[0], G.shape[1],
                                                        out.ctypes.data_as(C.c_void_p)))
        return out

    def time_function_sample(self, gamma, speed=1.0, t_cap=None, with_canonical=False):
        """(B, n_time_components) time latents -> (times (B, t_cap) float64 padded with NaN, lengths (B)): the spline's time function
        t'(t) of every row (mg_time_function_sample: back_project_time_function, motion_primitive.py:268-319, for the batch)."""
        G = _latents(gamma)
        B, F = G.shape[0], self.n_canonical_frames
        ctx = self.ctx
        cap = int(t_cap) if t_cap is not None else int(4 * F / min(float(speed), 1.0)) + 8
        while True:
            d_g, d_t, d_l = ctx.upload(G), ctx.malloc(max(B, 1) * cap * 8), ctx.malloc(max(B, 1) * 4)
            d_c = ctx.malloc(max(B, 1) * F * 8) if with_canonical else None
            try:
                _check(self.lib.mg_memset(ctx.handle, d_t.ptr, 0xff, max(B, 1) * cap * 8))      # NaN padding
                _check(self.lib.mg_time_function_sample(self.handle, d_g.ptr, _dtype_code(G), B, G.shape[1], float(speed), d_t.ptr, d_l.ptr, cap,
                                                        d_c.ptr if d_c is not None else None))
                lens = ctx.download(d_l, (B,), np.int32)
                if B and lens.min() < 0 and t_cap is None:      # a row needs more samples than assumed: once more with room for it
                    cap = int(-lens.min()) + 8
                    continue
                times = ctx.download(d_t, (B, cap), np.float64)
                canonical = ctx.download(d_c, (B, F), np.float64) if d_c is not None else None
            finally:
                for b in (d_g, d_t, d_l, d_c):
                    if b is not None:
                        b.free()
            return (times, lens, canonical) if with_canonical else (times, lens)

    def back_project_frames_at(self, S, times, lengths=None, dtype=np.float64):
        """(B, ld) latents, (B, t_cap) float64 times, (B) lengths -> (B, t_cap, D) frames of every candidate at ITS OWN times
        (mg_back_project_frames_at); samples beyond a row's length are NaN."""
        S = _latents(S)
        times = np.ascontiguousarray(times, dtype=np.float64)
        B, cap = times.shape
        if S.shape[0] != B:
            raise ValueError("one row of times per candidate")
        lens = None if lengths is None else np.ascontiguousarray(lengths, dtype=np.int32)
        item = np.dtype(dtype).itemsize
        ctx = self.ctx
        d_s, d_t, d_o = ctx.upload(S), ctx.upload(np.where(np.isnan(times), 0.0, times)), ctx.malloc(max(B, 1) * cap * self.n_dim * item)
        d_l = ctx.upload(lens) if lens is not None else None
        try:
            _check(self.lib.mg_memset(ctx.handle, d_o.ptr, 0xff, max(B, 1) * cap * self.n_dim * item))
            _check(self.lib.mg_back_project_frames_at(self.handle, d_s.ptr, _dtype_code(S), B, S.shape[1], d_t.ptr, d_l.ptr if d_l is not None else None, cap,
                                                      d_o.ptr, MG_F64 if np.dtype(dtype) == np.float64 else MG_F32))
            return ctx.download(d_o, (B, cap, self.n_dim), dtype)
        finally:
            for b in (d_s, d_t, d_o, d_l):
                if b is not None:
                    b.free()

    def time_grid(self, times):
        return TimeGrid(self, times)

    def _grid_handle(self, grid):
        return None if grid is None else grid.handle

    def _grid_size(self, grid):
        return self.canonical_grid.size if grid is None else grid.size

    # ---- host-array entry points (upload, launch, download) -------------------------------
    def back_project_frames(self, S, grid=None, path=MG_PATH_AUTO):
        S = _latents(S)
        out = np.empty((S.shape[0], self._grid_size(grid), self.n_dim), dtype=np.float32)
        _check(self.lib.mg_back_project_frames_host(self.handle, self._grid_handle(grid), S.ctypes.data_as(C.c_void_p),
                                                    _dtype_code(S), S.shape[0], S.shape[1],
                                                    out.ctypes.data_as(C.c_void_p), path))
        return out

    def back_project_frames_f64(self, S, grid=None):
        S = _latents(S)
        out = np.empty((S.shape[0], self._grid_size(grid), self.n_dim), dtype=np.float64)
        _check(self.lib.mg_back_project_frames_f64_host(self.handle, self._grid_handle(grid), S.ctypes.data_as(C.c_void_p),
                                                        _dtype_code(S), S.shape[0], S.shape[1],
                                                        out.ctypes.data_as(C.c_void_p)))
        return out

    def back_project_coeffs(self, S, dtype=np.float64):
        S = _latents(S)
        out = np.empty((S.shape[0], self.n_basis, self.n_dim), dtype=dtype)
        _check(self.lib.mg_back_project_coeffs_host(self.handle, S.ctypes.data_as(C.c_void_p), _dtype_code(S),
                                                    S.shape[0], S.shape[1], out.ctypes.data_as(C.c_void_p),
                                                    _dtype_code(out)))
        return out

    def spline_evaluate(self, coeffs, grid=None):
        c = np.ascontiguousarray(coeffs, dtype=np.float64)
        if c.ndim == 2:
            c = c.reshape(1, c.shape[0], c.shape[1])
        if c.shape[1:] != (self.n_basis, self.n_dim):
            raise ValueError("coeffs must be (n, %d, %d)" % (self.n_basis, self.n_dim))
        out = np.empty((c.shape[0], self._grid_size(grid), self.n_dim), dtype=np.float64)
        _check(self.lib.mg_spline_evaluate_host(self.handle, self._grid_handle(grid), c.ctypes.data_as(C.c_void_p),
                                                c.shape[0], out.ctypes.data_as(C.c_void_p)))
        return out

    def gmm_log_prob(self, X, dtype=np.float64):
        X = _latents(X)
        out = np.empty(X.shape[0], dtype=dtype)
        _check(self.lib.mg_gmm_log_prob_host(self.handle, X.ctypes.data_as(C.c_void_p), _dtype_code(X), X.shape[0],
                                             X.shape[1], out.ctypes.data_as(C.c_void_p), _dtype_code(out)))
        return out

    def gmm_sample(self, counts, seed, dtype=np.float64):
        counts = np.ascontiguousarray(counts, dtype=np.int64)
        n = int(counts.sum())
        X = np.empty((n, self.n_gmm_dims), dtype=dtype)
        comp = np.empty(n, dtype=np.int32)
        _check(self.lib.mg_gmm_sample_host(self.handle, n, counts.ctypes.data_as(C.c_void_p), C.c_uint64(int(seed)),
                                           X.ctypes.data_as(C.c_void_p), _dtype_code(X), self.n_gmm_dims,
                                           comp.ctypes.data_as(C.c_void_p)))
        return X, comp

    def score_constraints(self, cset, S, dtype=np.float64):
        S = _latents(S)
        out = np.empty(S.shape[0], dtype=dtype)
        _check(self.lib.mg_score_constraints_host(self.handle, cset.handle, S.ctypes.data_as(C.c_void_p), _dtype_code(S),
                                                  S.shape[0], S.shape[1], out.ctypes.data_as(C.c_void_p),
                                                  _dtype_code(out)))
        return out

    def best_candidate(self, cset, S):
        """(best_index, min_error) of the reference's candidate loop (first minimum) for host latents S."""
        S = _latents(S)
        idx, val = C.c_int64(), C.c_double()
        _check(self.lib.mg_best_candidate_host(self.handle, cset.handle, S.ctypes.data_as(C.c_void_p), _dtype_code(S),
                                               S.shape[0], S.shape[1], C.byref(idx), C.byref(val)))
        return idx.value, val.value

    def best_candidate_dev(self, cset, lat_dev, lat_dtype, n, ld):
        lc = MG_F64 if np.dtype(lat_dtype) == np.float64 else MG_F32
        idx, val = C.c_int64(), C.c_double()
        _check(self.lib.mg_best_candidate(self.handle, cset.handle, _dev_ptr(lat_dev), lc, int(n), int(ld),
                                          C.byref(idx), C.byref(val)))
        return idx.value, val.value

    def score_constraint_residuals(self, cset, S):
        """(n_samples, n_constraints) float64: weight_c * error_c per sample (get_residual_vector, batched)."""
        S = _latents(S)
        out = np.empty((S.shape[0], cset.n), dtype=np.float64)
        _check(self.lib.mg_score_constraint_residuals_host(self.handle, cset.handle, S.ctypes.data_as(C.c_void_p),
                                                           _dtype_code(S), S.shape[0], S.shape[1],
                                                           out.ctypes.data_as(C.c_void_p)))
        return out

    def score_constraint_residuals_chained(self, cset, S, align_cand):
        """The same matrix with every sample aligned to ITS OWN previous motion: align_cand (n, 4) = previous unit heading
        (x, z) and previous root position (x, z) per sample (mg_score_constraint_residuals_chained)."""
        S = _latents(S)
        A = np.ascontiguousarray(align_cand, dtype=np.float64)
        assert A.shape == (S.shape[0], 4)
        ctx = self.ctx
        d_S, d_A, d_R = ctx.upload(S), ctx.upload(A), ctx.malloc(max(S.shape[0] * cset.n, 1) * 8)
        try:
            _check(self.lib.mg_score_constraint_residuals_chained(self.handle, cset.handle, d_S.ptr, _dtype_code(S), S.shape[0], S.shape[1],
                                                                  d_A.ptr, d_R.ptr))
            return ctx.download(d_R, (S.shape[0], cset.n), np.float64)
        finally:
            for b in (d_S, d_A, d_R):
                b.free()

    def gmm_log_prob_jac(self, X):
        """(n_samples, n_components) float64: the reference's log_likelihood_jac (= -grad log p) per row."""
        X = _latents(X)
        out = np.empty((X.shape[0], self.n_gmm_dims), dtype=np.float64)
        _check(self.lib.mg_gmm_log_prob_jac_host(self.handle, X.ctypes.data_as(C.c_void_p), _dtype_code(X), X.shape[0],
                                                 X.shape[1], out.ctypes.data_as(C.c_void_p)))
        return out

    # ---- device-pointer entry points (no copies, asynchronous on the context stream) ----------
    def back_project_frames_dev(self, lat_dev, lat_dtype, n, ld, frames_dev, grid=None, path=MG_PATH_AUTO):
        code = MG_F64 if np.dtype(lat_dtype) == np.float64 else MG_F32
        _check(self.lib.mg_back_project_frames(self.handle, self._grid_handle(grid), _dev_ptr(lat_dev), code,
                                               int(n), int(ld), _dev_ptr(frames_dev), path))

    def gmm_log_prob_dev(self, x_dev, x_dtype, n, ld, out_dev, out_dtype=np.float32):
        xc = MG_F64 if np.dtype(x_dtype) == np.float64 else MG_F32
        oc = MG_F64 if np.dtype(out_dtype) == np.float64 else MG_F32
        _check(self.lib.mg_gmm_log_prob(self.handle, _dev_ptr(x_dev), xc, int(n), int(ld), _dev_ptr(out_dev), oc))

    def gmm_sample_dev(self, counts, seed, x_dev, x_dtype, ld, component_dev=None, rows=None):
        """Device Philox sampler into device memory: rows grouped by component like sklearn's sample().
        rows = (begin, count): only those rows of the draw (mg_gmm_sample_rows), x_dev (count, ld)."""
        counts = np.ascontiguousarray(counts, dtype=np.int64)
        xc = MG_F64 if np.dtype(x_dtype) == np.float64 else MG_F32
        comp = _dev_ptr(component_dev) if component_dev is not None else C.c_void_p(0)
        n = int(counts.sum())
        begin, count = (0, n) if rows is None else (int(rows[0]), int(rows[1]))
        _check(self.lib.mg_gmm_sample_rows(self.handle, n, counts.ctypes.data_as(C.c_void_p), C.c_uint64(int(seed)), begin, count,
                                           _dev_ptr(x_dev), xc, int(ld), comp))

    def objective_dev(self, cset, lat_dev, lat_dtype, n, ld, error_scale, quality_scale, obj_dev=None, err_dev=None, logp_dev=None):
        """error_scale * constraint error + quality_scale * (-log p) of n device-resident candidates in one launch
        (mg_objective_error_and_naturalness); outputs float64 device buffers, any of them None."""
        lc = MG_F64 if np.dtype(lat_dtype) == np.float64 else MG_F32
        ptr = lambda b: _dev_ptr(b) if b is not None else None
        _check(self.lib.mg_objective_error_and_naturalness(self.handle, cset.handle, _dev_ptr(lat_dev), lc, int(n), int(ld), float(error_scale),
                                                           float(quality_scale), ptr(logp_dev), ptr(err_dev), ptr(obj_dev)))

    def objective(self, cset, S, error_scale, quality_scale):
        """(objective, errors, log p) for host latents S (n, >= n_gmm_dims); raises MGError(-4) where the one-launch kernel does
        not carry the set."""
        S = _latents(S)
        n = S.shape[0]
        ctx = self.ctx
        d_S, bufs = ctx.upload(S), [ctx.malloc(max(n, 1) * 8) for _ in range(3)]
        try:
            self.objective_dev(cset, d_S, S.dtype, n, S.shape[1], error_scale, quality_scale, bufs[0], bufs[1], bufs[2])
            return tuple(ctx.download(b, (n,), np.float64) for b in bufs)
        finally:
            for b in [d_S] + bufs:
                b.free()

    def score_constraints_dev(self, cset, lat_dev, lat_dtype, n, ld, out_dev, out_dtype=np.float64):
        lc = MG_F64 if np.dtype(lat_dtype) == np.float64 else MG_F32
        oc = MG_F64 if np.dtype(out_dtype) == np.float64 else MG_F32
        _check(self.lib.mg_score_constraints(self.handle, cset.handle, _dev_ptr(lat_dev), lc, int(n), int(ld),
                                             _dev_ptr(out_dev), oc))

    FRAMES_KERNEL_NAMES = ("mg_frames_direct_kernel", "mg_frames_ws_kernel", "mg_frames_cs_kernel")

    def step_plan(self, n, frames_dev=None):
        """What step_frames_and_logp_dev launches for n candidates (into frames_dev, when given: slow-class pieces of the output
        arena get the tile-major kernel): dict(kernel, fused, workgroups, lds_bytes)."""
        plan = (C.c_int32 * 4)()
        _check(self.lib.mg_step_plan_for(self.handle, int(n), _dev_ptr(frames_dev) if frames_dev is not None else None, plan))
        return dict(kernel=self.FRAMES_KERNEL_NAMES[plan[0]], fused=bool(plan[1]), workgroups=int(plan[2]), lds_bytes=int(plan[3]))

    def step_frames_and_logp_dev(self, lat_dev, lat_dtype, n, ld, frames_dev, logp_dev):
        code = MG_F64 if np.dtype(lat_dtype) == np.float64 else MG_F32
        _check(self.lib.mg_step_frames_and_logp(self.handle, _dev_ptr(lat_dev), code, int(n), int(ld),
                                                _dev_ptr(frames_dev), _dev_ptr(logp_dev)))
