"""ORACLE (test infrastructure, NOT product code) -- NumPy restatement of the
morphablegraphs motion-primitive hot path.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this module; nothing under morphablegraphs_amd/ does.  It is the CPU checker the
HIP path is compared against, never the thing shipped or measured.

Parity status: PINNED.  The reference holds no tests or golden vectors for this
path, so the restatement is pinned against outputs of the
reference's own, unmodified motion_primitive.py / motion_spline.py imported in
the build container (oracle/gen_golden.py -> tests/golden/*.npz) and is checked
against those fixtures by tests/test_oracle_golden.py.

Each function cites the reference file:line (relative to /root/reference) it
restates.  Arithmetic that lives in third-party code is restated from its
published algorithm:
  * scipy.interpolate.splev  = FITPACK splev.f / fpbspl.f (reference pins
    scipy==1.2.1; call sites motion_spline.py:86,92)
  * sklearn GaussianMixture.score_samples / sample / _compute_precision_cholesky
    (reference pins scikit_learn==0.23.2; call sites motion_primitive.py:136-144,189)
"""
import math
import numpy as np

B_SPLINE_DEGREE = 3  # morphablegraphs/motion_model/__init__.py:7


# --------------------------------------------------------------------------
# knots and FITPACK B-spline evaluation
# --------------------------------------------------------------------------
def cubic_b_spline_knots(n_basis, n_canonical_frames):
    """morphablegraphs/construction/utils.py:187-198 (get_cubic_b_spline_knots)."""
    knots = np.zeros(4 + n_basis)
    knots[3:-3] = np.linspace(0, n_canonical_frames - 1, n_basis - 2)
    knots[-3:] = n_canonical_frames - 1
    return knots


def fitpack_find_span(t, k, x):
    """FITPACK splev.f interval search: returns 0-based l with t[l] <= x < t[l+1],
    clamped to [k, n-k-2].  Points outside [t[k], t[n-k-1]] keep the first/last
    interval, i.e. ext=0 extrapolates with the end polynomial (the only mode
    motion_spline.py:86,92 uses)."""
    n = len(t)
    l = k
    while not (x < t[l + 1] or l == n - k - 2):
        l += 1
    return l


def fitpack_basis(t, k, x, l):
    """FITPACK fpbspl.f: the k+1 non-zero B-splines of degree k at x on span l
    (0-based), evaluated with the de Boor-Cox recurrence in fpbspl's operation order."""
    h = [0.0] * (k + 2)
    hh = [0.0] * (k + 1)
    h[0] = 1.0
    for j in range(1, k + 1):
        for i in range(j):
            hh[i] = h[i]
        h[0] = 0.0
        for i in range(1, j + 1):
            li = l + i
            lj = li - j
            if t[li] == t[lj]:
                h[i] = 0.0
                continue
            f = hh[i - 1] / (t[li] - t[lj])
            h[i - 1] = h[i - 1] + f * (t[li] - x)
            h[i] = f * (x - t[lj])
    return h[:k + 1]


def basis_rows(knots, time_points, k=B_SPLINE_DEGREE):
    """For every time point: first contributing coefficient index i0 (= l-k) and
    the k+1 weights.  frames[f, d] = sum_j w[f, j] * coeffs[i0[f] + j, d]."""
    knots = np.asarray(knots, dtype=np.float64)
    tp = np.atleast_1d(np.asarray(time_points, dtype=np.float64))
    i0 = np.empty(len(tp), dtype=np.int64)
    w = np.empty((len(tp), k + 1), dtype=np.float64)
    for f, x in enumerate(tp):
        l = fitpack_find_span(knots, k, x)
        i0[f] = l - k
        w[f] = fitpack_basis(knots, k, x, l)
    return i0, w


def splev(time_points, knots, coeffs_1d, k=B_SPLINE_DEGREE):
    """scipy.interpolate.splev(x, (t, c, k)) restated (ext=0)."""
    i0, w = basis_rows(knots, time_points, k)
    c = np.asarray(coeffs_1d, dtype=np.float64)
    out = np.zeros(len(i0))
    for j in range(k + 1):          # splev.f: sp = sp + c(ll)*h(j), j ascending
        out = out + c[i0 + j] * w[:, j]
    return out


def spline_frames(knots, coeffs, time_points, k=B_SPLINE_DEGREE):
    """motion_spline.py:71-92: all channels of coeffs (NB, D) at time_points -> (T, D)."""
    i0, w = basis_rows(knots, time_points, k)
    coeffs = np.asarray(coeffs, dtype=np.float64)
    out = np.zeros((len(i0), coeffs.shape[1]))
    for j in range(k + 1):
        out = out + coeffs[i0 + j, :] * w[:, j:j + 1]
    return out


# --------------------------------------------------------------------------
# Gaussian mixture (sklearn restated)
# --------------------------------------------------------------------------
def precision_cholesky(covars):
    """sklearn _compute_precision_cholesky(covariance_type='full'), used at
    motion_primitive.py:141-142 and motion_primitive_wrapper.py:366:
    P_k = solve_triangular(cholesky(S_k, lower), I, lower).T  (upper triangular)."""
    covars = np.asarray(covars, dtype=np.float64)
    K, L, _ = covars.shape
    out = np.empty_like(covars)
    for k in range(K):
        c = np.linalg.cholesky(covars[k])
        # forward substitution, column by column of the identity
        inv = np.zeros((L, L))
        for col in range(L):
            for r in range(col, L):
                s = 1.0 if r == col else 0.0
                s -= np.dot(c[r, col:r], inv[col:r, col])
                inv[r, col] = s / c[r, r]
        out[k] = inv.T
    return out


def gmm_component_log_prob(X, weights, means, prec_chol):
    """Weighted per-component log densities, (B, K):
    log w_k - 0.5 (L log 2pi + ||(x-mu_k) P_k||^2) + sum log diag P_k
    (sklearn _estimate_log_gaussian_prob + _estimate_log_weights; formula twin at
    morphablegraphs/motion_model/extended_mgrd_mixture_model.py:60-108)."""
    X = np.atleast_2d(np.asarray(X, dtype=np.float64))
    K, L = means.shape
    out = np.empty((X.shape[0], K))
    for k in range(K):
        y = (X - means[k]) @ prec_chol[k]
        log_det = np.sum(np.log(np.diagonal(prec_chol[k])))
        out[:, k] = -0.5 * (L * math.log(2.0 * math.pi) + np.sum(y * y, axis=1)) + log_det + math.log(weights[k])
    return out


def logsumexp_rows(a):
    """extended_mgrd_mixture_model.py:85-96 (max-shifted log-sum-exp)."""
    vmax = a.max(axis=1)
    return np.log(np.sum(np.exp(a - vmax[:, None]), axis=1)) + vmax


def gmm_log_prob(X, weights, means, prec_chol):
    """Per-sample log p(x) == sklearn GaussianMixture.score_samples(X)."""
    return logsumexp_rows(gmm_component_log_prob(X, weights, means, prec_chol))


def gmm_log_likelihood_jac(s, weights, means, covars, prec_chol):
    """log_likelihood_jac restated line by line
    (reference morphablegraphs/motion_generator/optimization/objective_functions.py:95-107; the same code is
    inlined in obj_spatial_error_sum_and_naturalness_jac, :190-206):
        numerator = sum_i exp(logN_i(s)) * w_i * inv(cov_i) @ (s - mean_i);  denominator = exp(score(s))
        return numerator / denominator  if denominator != 0 else ones
    i.e. MINUS the gradient of log p(s).  PARITY UNPINNED against the reference's own run: its module imports
    anim_utils (absent) and reads pre-0.18 sklearn attribute names; pinned instead by the finite-difference
    identity jac == -d/ds score_samples(s) on the golden mixtures (tests/test_oracle_golden.py)."""
    s = np.asarray(s, dtype=np.float64)
    K, L = means.shape
    numerator = np.zeros(L)
    for i in range(K):
        y = (s - means[i]) @ prec_chol[i]
        log_n = -0.5 * (L * math.log(2.0 * math.pi) + np.dot(y, y)) + np.sum(np.log(np.diagonal(prec_chol[i])))
        numerator += np.exp(log_n) * weights[i] * np.dot(np.linalg.inv(covars[i]), (s - means[i]))
    denominator = np.exp(gmm_log_prob(s[None, :], weights, means, prec_chol)[0])
    if denominator != 0:
        return numerator / denominator
    return np.ones(s.shape)


def gmm_sample(n_samples, weights, means, covars, rng):
    """sklearn GaussianMixture.sample restated (motion_primitive.py:189):
    counts = multinomial(n, w); rows grouped by component, not shuffled."""
    counts = rng.multinomial(n_samples, weights)
    X = np.vstack([rng.multivariate_normal(m, c, int(cnt)) for m, c, cnt in zip(means, covars, counts)])
    y = np.concatenate([np.full(int(cnt), j, dtype=int) for j, cnt in enumerate(counts)])
    return X, y


# --------------------------------------------------------------------------
# candidate scoring
# --------------------------------------------------------------------------
def first_min_argmin(errors):
    """motion_primitive_generator.py:251-257: strict '>' keeps the FIRST minimum;
    returns (0, inf) for an empty list, NaN errors never win."""
    best_idx, min_error = 0, np.inf
    for idx, e in enumerate(errors):
        if min_error > e:
            min_error, best_idx = e, idx
    return best_idx, min_error


def point_distance(target, p):
    """global_transform_constraint.py:135-143: distance ignoring None axes."""
    d = 0.0
    for i in range(3):
        if target[i] is not None and not (isinstance(target[i], float) and math.isnan(target[i])):
            d += (target[i] - p[i]) ** 2
    return math.sqrt(d)


def quaternion_rotate(q, v):
    """Rotate v by unit-or-not quaternion q = (w, x, y, z) through the rotation
    matrix of q (as transformations.quaternion_matrix does: normalises by |q|^2)."""
    w, x, y, z = [float(c) for c in q]
    n = w * w + x * x + y * y + z * z
    s = 2.0 / n
    m = np.array([[1 - s * (y * y + z * z), s * (x * y - z * w), s * (x * z + y * w)],
                  [s * (x * y + z * w), 1 - s * (x * x + z * z), s * (y * z - x * w)],
                  [s * (x * z - y * w), s * (y * z + x * w), 1 - s * (x * x + y * y)]])
    return m @ np.asarray(v, dtype=np.float64)


def direction_2d_error(target_dir_xz, root_quat, ref_dir=(0.0, 0.0, 1.0)):
    """direction_2d_constraint.py:42-52 with the anim_utils FK call replaced by its
    root-joint special case (PARITY UNPINNED: anim_utils source is not under
    /root/reference).  Heading = xz of the root rotation applied to ref_dir."""
    t = np.asarray(target_dir_xz, dtype=np.float64)
    t = t / np.linalg.norm(t)
    p = quaternion_rotate(root_quat, ref_dir)
    m = np.array([p[0], p[2]])
    m = m / np.linalg.norm(m)
    cos_angle = float(np.dot(t, m) / (np.linalg.norm(t) * np.linalg.norm(m)))
    cos_angle = min(1.0, max(cos_angle, -1.0))
    return abs(math.degrees(math.acos(cos_angle)))


def quaternion_matrix3(q):
    """Rotation matrix of a (w, x, y, z) quaternion, normalising like transformations.quaternion_matrix
    (which the reference's FK reaches through anim_utils): q *= sqrt(2 / dot(q, q)), outer products."""
    q = np.asarray(q, dtype=np.float64)
    n = np.dot(q, q)
    q = q * math.sqrt(2.0 / n)
    o = np.outer(q, q)
    return np.array([[1.0 - o[2, 2] - o[3, 3], o[1, 2] - o[3, 0], o[1, 3] + o[2, 0]],
                     [o[1, 2] + o[3, 0], 1.0 - o[1, 1] - o[3, 3], o[2, 3] - o[1, 0]],
                     [o[1, 3] - o[2, 0], o[2, 3] + o[1, 0], 1.0 - o[1, 1] - o[2, 2]]])


def joint_global_position(frame, joints, animated_joints, joint):
    """SELF-DEFINED forward kinematics (the reference calls anim_utils' SkeletonNode.get_global_position, absent
    here; PARITY UNPINNED): walk the chain root -> joint with 3x3 matrices, p += R_parent_global @ offset.
    frame: root translation [0:3] then one (w,x,y,z) per animated joint; joints: [(name, parent, offset)]."""
    by_name = {j[0]: j for j in joints}
    chan = {n: 3 + 4 * i for i, n in enumerate(animated_joints)}
    chain, n = [], joint
    while n is not None:
        chain.insert(0, n)
        n = by_name[n][1]
    p = np.array(frame[:3], dtype=np.float64)
    R = np.eye(3)
    for parent, child in zip(chain[:-1], chain[1:]):
        if parent in chan:
            R = R @ quaternion_matrix3(frame[chan[parent]:chan[parent] + 4])
        p = p + R @ np.asarray(by_name[child][2], dtype=np.float64)
    return p


def joint_global_orientation(frame, joints, animated_joints, joint):
    """3x3 global orientation of `joint` (its own rotation included) by chaining rotation matrices root -> joint."""
    by_name = {j[0]: j for j in joints}
    chan = {n: 3 + 4 * i for i, n in enumerate(animated_joints)}
    chain, n = [], joint
    while n is not None:
        chain.insert(0, n)
        n = by_name[n][1]
    R = np.eye(3)
    for name in chain:
        if name in chan:
            R = R @ quaternion_matrix3(frame[chan[name]:chan[name] + 4])
    return R


def joint_orientation_error(frame, joints, animated_joints, joint, target_quat, ref_dir=(0.0, 0.0, 1.0)):
    """GlobalTransformConstraint._quaternion_distance (global_transform_constraint.py:109-121): the joint's global
    orientation and the wanted one both applied to the z axis (ORIGIN = [0,0,0,1] read as the pure quaternion of
    (0,0,1)), then transformations.angle_between_vectors = arccos(v1 . v2 / (|v1| |v2|)) in radians.  SELF-DEFINED
    for the anim_utils pieces (get_global_orientation_quaternion, quaternion_rotate_vector); the dot product is
    clamped to [-1, 1] where the reference would return NaN by rounding."""
    v1 = joint_global_orientation(frame, joints, animated_joints, joint) @ np.asarray(ref_dir, dtype=np.float64)
    v2 = quaternion_matrix3(target_quat) @ np.asarray(ref_dir, dtype=np.float64)
    d = float(np.dot(v1, v2) / (np.linalg.norm(v1) * np.linalg.norm(v2)))
    return math.acos(min(1.0, max(d, -1.0)))


def two_hand_residuals(frame, joints, animated_joints, joint_names, positions):
    """TwoHandConstraint.get_residual_vector_frame (two_hand_constraint.py:66-74): distance of the hands' centre to
    the targets' centre, left hand to its target, right hand to its target."""
    left = joint_global_position(frame, joints, animated_joints, joint_names[0])
    right = joint_global_position(frame, joints, animated_joints, joint_names[1])
    p0, p1 = np.asarray(positions[0], dtype=np.float64), np.asarray(positions[1], dtype=np.float64)
    delta = right - left
    center = p0 + 0.5 * (p1 - p0)
    return [float(np.linalg.norm(center - (left + 0.5 * delta))), float(np.linalg.norm(p0 - left)), float(np.linalg.norm(p1 - right))]


def feet_residuals(frame, joints, animated_joints, left, right, weight_factor=1.0):
    """FeetConstraint.get_residual_vector (feet_constraint.py:47-51): distance of each foot to its target, the constraint's own
    weight factor applied INSIDE (the caller multiplies once more); get_residual_vector_spline is the ONE entry [left + right]."""
    lp = joint_global_position(frame, joints, animated_joints, "LeftFoot")
    rp = joint_global_position(frame, joints, animated_joints, "RightFoot")
    return [float(np.linalg.norm(np.asarray(left, dtype=np.float64) - lp)) * weight_factor,
            float(np.linalg.norm(np.asarray(right, dtype=np.float64) - rp)) * weight_factor]


def time_constraints_start_keyframe(time_functions_before):
    """TimeConstraints._get_start_frame (time_constraints.py:32-39): the last entries of the time functions of the steps before
    start_step, summed."""
    start = 0
    for tf in time_functions_before:
        start += tf[-1]
    return start


def time_constraints_error(time_functions, constraint_list, start_keyframe, frame_time):
    """TimeConstraints.evaluate_graph_walk / calculate_constraint_error (time_constraints.py:40-50, 68-91) on the time functions
    of the steps start_step .. end_step: frames counted up to the constrained step's keyframe (int(t[key]) + 1 of them in that
    step), squared difference to the desired time; 0 when the keyframe is beyond the step's time function, 10000 when the
    constrained step is beyond the walk."""
    error_sum = 0.0
    for step_index, keyframe_index, desired_time in constraint_list:
        n_frames, error = start_keyframe, 10000.0
        for k, tf in enumerate(time_functions):
            if k < int(step_index):
                n_frames += tf[-1]
                continue
            if int(keyframe_index) >= len(tf):
                error = 0.0
            else:
                n_frames += int(tf[int(keyframe_index)]) + 1
                error = (desired_time - n_frames * frame_time) ** 2
            break
        error_sum += error
    return error_sum


def node_heading(frame, joints, animated_joints, joint, ref_dir=(0.0, 0.0, 1.0)):
    """What anim_utils' get_global_node_orientation_vector is documented to return: unit (x, z) of the node's global
    rotation applied to ref_dir (SELF-DEFINED, anim_utils absent)."""
    p = joint_global_orientation(frame, joints, animated_joints, joint) @ np.asarray(ref_dir, dtype=np.float64)
    d = np.array([p[0], p[2]])
    return d / np.linalg.norm(d)


def quaternion_multiply(a, b):
    """Hamilton product of (w, x, y, z) quaternions (transformations.quaternion_multiply)."""
    w0, x0, y0, z0 = a
    w1, x1, y1, z1 = b
    return np.array([w0 * w1 - x0 * x1 - y0 * y1 - z0 * z1, w0 * x1 + x0 * w1 + y0 * z1 - z0 * y1,
                     w0 * y1 - x0 * z1 + y0 * w1 + z0 * x1, w0 * z1 + x0 * y1 - y0 * x1 + z0 * w1])


def align_coeffs_to_previous_frame(coeffs, prev_frame, joints, animated_joints, joint, ref_dir=(0.0, 0.0, 1.0)):
    """SELF-DEFINED restatement of what motion_primitive_constraints.py:110-114 asks of anim_utils'
    align_quaternion_frames_automatically (absent here; PARITY UNPINNED), written the way that library works -- on
    the control points, with a 4x4 matrix and a quaternion: angle between the aligning node's heading in the last
    previous frame and in the FIRST control point, rotation about y by it, translation in x and z that puts the
    first root position on the previous one; every control point's root position goes through the matrix and its
    root quaternion is multiplied by the rotation from the left.  Returns a new (n_basis, D) array."""
    coeffs = np.array(coeffs, dtype=np.float64)
    ha = node_heading(prev_frame, joints, animated_joints, joint, ref_dir)
    hb = node_heading(coeffs[0], joints, animated_joints, joint, ref_dir)
    # a rotation about +y by phi turns the xz heading angle atan2(z, x) by -phi
    phi = math.atan2(hb[1], hb[0]) - math.atan2(ha[1], ha[0])
    q = np.array([math.cos(phi / 2.0), 0.0, math.sin(phi / 2.0), 0.0])
    m = np.eye(4)
    m[:3, :3] = quaternion_matrix3(q)
    rotated_first = m @ np.array([coeffs[0][0], coeffs[0][1], coeffs[0][2], 1.0])
    m[0, 3] = prev_frame[0] - rotated_first[0]
    m[2, 3] = prev_frame[2] - rotated_first[2]
    for cp in coeffs:
        cp[:3] = (m @ np.array([cp[0], cp[1], cp[2], 1.0]))[:3]
        cp[3:7] = quaternion_multiply(q, cp[3:7])
    return coeffs


def align_coeffs_to_start_pose(coeffs, start_pose):
    """The other branch of align_quaternion_frames (optimization/objective_functions.py:38-47), taken when there are no
    previous frames but a start pose: m = get_transform_from_start_pose(start_pose) (anim_utils, absent: PARITY UNPINNED;
    restated as the rotation by the start orientation's Euler angles in degrees plus the start position, here for
    rotations about y only); t_pos = m . (first root position, 1); delta = start_pose["position"] -- the SAME list
    object, so the two subtractions that follow also rewrite the start pose for the next call; m[:3, 3] = delta; every
    control point goes through m (positions through the matrix, root quaternions multiplied by its rotation from the
    left).  Returns a new (n_basis, D) array and mutates start_pose["position"] exactly as the reference does."""
    coeffs = np.array(coeffs, dtype=np.float64)
    rx, ry, rz = [math.radians(float(v)) for v in start_pose["orientation"]]
    assert rx == 0.0 and rz == 0.0, "rotations about y only"
    q = np.array([math.cos(ry / 2.0), 0.0, math.sin(ry / 2.0), 0.0])
    m = np.eye(4)
    m[:3, :3] = quaternion_matrix3(q)
    m[:3, 3] = start_pose["position"]
    first_frame_pos = coeffs[0][:3].tolist() + [1]
    t_pos = np.dot(m, first_frame_pos)[:3]
    delta = start_pose["position"]
    delta[0] -= t_pos[0]
    delta[2] -= t_pos[2]
    m[:3, 3] = delta
    for cp in coeffs:
        cp[:3] = (m @ np.array([cp[0], cp[1], cp[2], 1.0]))[:3]
        cp[3:7] = quaternion_multiply(q, cp[3:7])
    return coeffs


# --------------------------------------------------------------------------
# trajectory constraints (root path against a Catmull-Rom spline)
# --------------------------------------------------------------------------
_CATMULL_ROM_BASE = np.array([[-1.0, 3.0, -3.0, 1.0], [2.0, -5.0, 4.0, -1.0], [-1.0, 0.0, 1.0, 0.0], [0.0, 2.0, 0.0, 0.0]])


def catmull_rom_point(control_points, u):
    """CatmullRomSpline.query_point_by_parameter (constraints/spatial_constraints/splines/catmull_rom_spline.py:66-71,
    118-168): the control points padded to [P0] + P + [Pn, Pn]; segment = min(floor(N u), N) + 1 with N = len(P) - 1,
    local parameter = N u - floor; 0.5 * [t^3 t^2 t 1] . base . [P_(i-1) P_i P_(i+1) P_(i+2)]; past the last segment the
    last control point.  PINNED by tests/golden/trajectory_spline.npz (points and arc length made by the reference)."""
    P = [np.asarray(p, dtype=np.float64) for p in control_points]
    n_seg = len(P) - 1
    padded = [P[0]] + P + [P[-1], P[-1]]
    scaled = n_seg * float(u)
    index = min(int(math.floor(scaled)), n_seg)
    t = scaled - index
    seg = index + 1
    if seg > n_seg:
        return padded[-1].copy()
    w = np.array([t ** 3, t ** 2, t, 1.0])
    ctrl = np.stack([padded[seg - 1], padded[seg], padded[seg + 1], padded[seg + 2]])      # (4, dims)
    return 0.5 * (w @ (_CATMULL_ROM_BASE @ ctrl))


def catmull_rom_derivatives(control_points, u):
    """dP/du and d2P/du2 of the same curve (the segment's cubic differentiated; zero past the last segment)."""
    P = [np.asarray(p, dtype=np.float64) for p in control_points]
    n_seg = len(P) - 1
    padded = [P[0]] + P + [P[-1], P[-1]]
    scaled = n_seg * float(u)
    index = min(int(math.floor(scaled)), n_seg)
    t = scaled - index
    seg = index + 1
    if seg > n_seg:
        return np.zeros_like(P[0]), np.zeros_like(P[0])
    ctrl = np.stack([padded[seg - 1], padded[seg], padded[seg + 1], padded[seg + 2]])
    A = 0.5 * (_CATMULL_ROM_BASE @ ctrl)                    # rows: t^3, t^2, t, 1
    d1 = (3.0 * A[0] * t + 2.0 * A[1]) * t + A[2]
    d2 = 6.0 * A[0] * t + 2.0 * A[1]
    return n_seg * d1, n_seg * n_seg * d2


def catmull_rom_full_arc_length(control_points, granularity=1000):
    """RelativeArcLengthMap._update_table (splines/arc_length_map.py:45-71): polyline length over granularity + 1 samples."""
    pts = np.array([catmull_rom_point(control_points, k / float(granularity)) for k in range(granularity + 1)])
    return float(np.sum(np.linalg.norm(pts[1:] - pts[:-1], axis=1)))


def closest_point_from(control_points, point, min_u):
    """ParameterizedSpline.find_closest_point_fast (splines/parameterized_spline.py:303-322): L-BFGS-B on the distance over
    the spline parameter, bounds [min_u, 1], started AT min_u -- the reference's call itself, through the installed scipy, with the
    parameter unwrapped to a float (under NumPy >= 1.24 the reference's weight vector [x**3, x**2, x, 1] is ragged when scipy hands
    over x as a 1-element array).  PINNED by tests/golden/trajectory_closest_point.npz (the reference's own run: 2e-7)."""
    from scipy.optimize import minimize
    target = np.asarray(point, dtype=np.float64)

    def dist(x):
        return float(np.linalg.norm(catmull_rom_point(control_points, float(np.ravel(x)[0])) - target))
    res = minimize(dist, np.array([float(min_u)]), method="L-BFGS-B", bounds=[(float(min_u), 1.0)])
    u = float(res["x"][0])
    return catmull_rom_point(control_points, u), u


def trajectory_residuals(root_path, control_points, min_u=0.0):
    """TrajectoryConstraint.get_residual_vector for the root joint (constraints/spatial_constraints/trajectory_constraint.py:
    95-121): per frame the distance to the closest spline point at or after the previous frame's parameter."""
    errors = np.empty(len(root_path))
    for f, p in enumerate(root_path):
        target, u = closest_point_from(control_points, p, min_u)
        errors[f] = np.linalg.norm(np.asarray(p, dtype=np.float64) - target)
        min_u = u
    return errors


# ---- scipy's L-BFGS-B for ONE bounded variable, restated -------------------------------------------------------------
# ParameterizedSpline.find_closest_point_fast is scipy.optimize.minimize(method="L-BFGS-B", bounds=[(min_u, 1)]) with a
# forward-difference gradient: not a local search -- its first step goes to the far bound and the More'-Thuente line search
# interpolates back from there, so WHICH local minimum of the distance a frame lands in is decided by that algorithm's own
# arithmetic (measured on tests/golden/trajectory_closest_point.npz: a monotone local walk ends up to 0.9 of the parameter range
# away from the reference on 12 of 26 tracks).  Results identical to the reference's therefore need the algorithm itself:
# L-BFGS-B 3.0 (Byrd, Lu, Nocedal, Zhu 1995; Morales, Nocedal 2011; the reference pins scipy 1.2.1 -- Fortran --, 1.15.3 -- the C
# translation -- is installed and scipy 1.7.1 -- Fortran -- sits in the image's conda tree: both behave alike on every case tried)
# specialised to n = 1:
#   * the limited-memory matrix collapses to the scalar theta = y'y / s'y = y / s (any BFGS update in one dimension gives
#     B = y / s, whatever the history), so the generalised Cauchy point is clamp(x - g / theta) and, where it is interior, it IS the
#     model's minimiser: the subspace minimisation (subsm) moves it by rounding noise only and is left out;
#   * what cannot be left out is formk's BOOKKEEPING: the routine that prepares subsm keeps the matrix WN1 up to date
#     incrementally (a new row per accepted pair, corrections when a variable enters or leaves the free set) and is only called in
#     iterations whose Cauchy point is free; an iteration whose Cauchy point sits on a bound skips it, its pair's row is never
#     written and the "leaves the free set" correction never applied -- the next call then factorises a wrong matrix, its
#     Cholesky step usually fails ("nonpositive definiteness in formk; refresh the lbfgs memory and restart the iteration"), and
#     the restart throws the curvature away (theta = 1: the next step goes to a bound again).  With one variable that happens in
#     about one search in eight, so _FormK below transcribes the bookkeeping and the two factorizations for n = 1.
# Third-party algorithm, published: routines mainlb, projgr, cauchy, freev, formk, matupd, lnsrlb, dpofa of L-BFGS-B 3.0 / LINPACK,
# dcsrch, dcstep of MINPACK-2; scipy's wrapper: optimize/_lbfgsb_py.py (_minimize_lbfgsb), optimize/_numdiff.py
# (approx_derivative, 2-point, abs_step = 1e-8, _adjust_scheme_to_bounds), optimize/_minimize.py
# (_optimize_result_for_equal_bounds).  PINNED: against scipy itself on random problems (tests/test_oracle_golden.py: parameter,
# iteration count, exit reason) and, chained, against tests/golden/trajectory_closest_point.npz, which the reference's own
# find_closest_point_fast produced.
_LB_EPSMCH = 2.220446049250313e-16
_LB_M = 10


class _FormK(object):
    """formk's state for one variable: WN1's three blocks (1-based, m = 10), and the two Cholesky steps' verdict."""

    def __init__(self):
        m = _LB_M
        self.A = [[0.0] * (m + 1) for _ in range(m + 1)]      # block (1,1): Y'ZZ'Y      (lower triangle used)
        self.C = [[0.0] * (m + 1) for _ in range(m + 1)]      # block (2,2): S'AA'S      (lower triangle used)
        self.Bm = [[0.0] * (m + 1) for _ in range(m + 1)]     # block (2,1): L_a + R_z   (full)

    def call(self, free, entered, left, updatd, iupdat, col, head, ws, wy, dr, theta):
        """One call of formk (the variable is `free` at the Cauchy point; it `entered` / `left` the free set since the last freev).
        ws, wy, dr: the pairs' s, y and s'y by ring position (1-based).  True when both factorizations succeed."""
        m, A, C, Bm = _LB_M, self.A, self.C, self.Bm
        if updatd:
            if iupdat > m:                         # shift old part of WN1
                for jy in range(1, m):
                    for i in range(m - jy):
                        A[jy + i][jy] = A[jy + 1 + i][jy + 1]
                        C[jy + i][jy] = C[jy + 1 + i][jy + 1]
                    for i in range(m - 1):
                        Bm[1 + i][jy] = Bm[2 + i][jy + 1]
            ipntr = head + col - 1
            if ipntr > m:
                ipntr -= m
            jpntr = head
            for jy in range(1, col + 1):           # new rows in blocks (1,1), (2,1) and (2,2)
                A[col][jy] = wy[ipntr] * wy[jpntr] if free else 0.0
                C[col][jy] = 0.0 if free else ws[ipntr] * ws[jpntr]
                Bm[col][jy] = 0.0 if free else ws[ipntr] * wy[jpntr]
                jpntr = jpntr % m + 1
            jpntr = head + col - 1
            if jpntr > m:
                jpntr -= m
            ipntr = head
            for i in range(1, col + 1):            # new column in block (2,1)
                Bm[i][col] = ws[ipntr] * wy[jpntr] if free else 0.0
                ipntr = ipntr % m + 1
            upcl = col - 1
        else:
            upcl = col
        ipntr = head
        for iy in range(1, upcl + 1):              # the old parts of (1,1) and (2,2) follow the free set
            jpntr = head
            for jy in range(1, iy + 1):
                t1 = wy[ipntr] * wy[jpntr] if entered else 0.0
                t2 = ws[ipntr] * ws[jpntr] if entered else 0.0
                t3 = wy[ipntr] * wy[jpntr] if left else 0.0
                t4 = ws[ipntr] * ws[jpntr] if left else 0.0
                A[iy][jy] = A[iy][jy] + t1 - t3
                C[iy][jy] = C[iy][jy] - t2 + t4
                jpntr = jpntr % m + 1
            ipntr = ipntr % m + 1
        ipntr = head
        for i_s in range(1, upcl + 1):             # ... and of (2,1)
            jpntr = head
            for jy in range(1, upcl + 1):
                t1 = ws[ipntr] * wy[jpntr] if entered else 0.0
                t3 = ws[ipntr] * wy[jpntr] if left else 0.0
                if i_s <= jy:
                    Bm[i_s][jy] = Bm[i_s][jy] + t1 - t3
                else:
                    Bm[i_s][jy] = Bm[i_s][jy] - t1 + t3
                jpntr = jpntr % m + 1
            ipntr = ipntr % m + 1
        # the upper triangle of WN = [D + Y'ZZ'Y / theta, -L_a' + R_z'; ., S'AA'S theta]
        n2 = 2 * col
        wn = [[0.0] * (n2 + 1) for _ in range(n2 + 1)]
        for iy in range(1, col + 1):
            i_s = col + iy
            for jy in range(1, iy + 1):
                wn[jy][iy] = A[iy][jy] / theta
                wn[col + jy][i_s] = C[iy][jy] * theta
            for jy in range(1, iy):
                wn[jy][i_s] = -Bm[iy][jy]
            for jy in range(iy, col + 1):
                wn[jy][i_s] = Bm[iy][jy]
            pos = head + iy - 1
            if pos > m:
                pos -= m
            wn[iy][iy] = wn[iy][iy] + dr[pos]
        if not _dpofa(wn, 0, col):
            return False
        for js in range(col + 1, n2 + 1):          # L^-1 (-L_a' + R_z') in the (1,2) block: solve trans(L') x = b
            for j in range(1, col + 1):
                if wn[j][j] == 0.0:
                    return False
            wn[1][js] = wn[1][js] / wn[1][1]
            for j in range(2, col + 1):
                acc = 0.0
                for k in range(1, j):
                    acc += wn[k][j] * wn[k][js]
                wn[j][js] = (wn[j][js] - acc) / wn[j][j]
        for i_s in range(col + 1, n2 + 1):
            for js in range(i_s, n2 + 1):
                acc = 0.0
                for k in range(1, col + 1):
                    acc += wn[k][i_s] * wn[k][js]
                wn[i_s][js] = wn[i_s][js] + acc
        return _dpofa(wn, col, col)


def _dpofa(a, off, n):
    """LINPACK dpofa on the upper triangle of a[off+1 .. off+n][off+1 .. off+n]: False where a pivot is not positive."""
    for j in range(1, n + 1):
        sacc = 0.0
        for k in range(1, j):
            dot = 0.0
            for i in range(1, k):
                dot += a[off + i][off + k] * a[off + i][off + j]
            t = a[off + k][off + j] - dot
            t = t / a[off + k][off + k]
            a[off + k][off + j] = t
            sacc += t * t
        sacc = a[off + j][off + j] - sacc
        if sacc <= 0.0:
            return False
        a[off + j][off + j] = math.sqrt(sacc)
    return True


def _dcstep(stx, fx, dx, sty, fy, dy, stp, fp, dp, brackt, stpmin, stpmax):
    """MINPACK-2 dcstep: the safeguarded cubic / quadratic step and the update of the interval of uncertainty."""
    sgnd = dp * (dx / abs(dx))
    if fp > fx:
        theta = 3.0 * (fx - fp) / (stp - stx) + dx + dp
        sc = max(abs(theta), abs(dx), abs(dp))
        gamma = sc * math.sqrt((theta / sc) * (theta / sc) - (dx / sc) * (dp / sc))
        if stp < stx:
            gamma = -gamma
        p = (gamma - dx) + theta
        q = ((gamma - dx) + gamma) + dp
        r = p / q
        stpc = stx + r * (stp - stx)
        stpq = stx + ((dx / ((fx - fp) / (stp - stx) + dx)) / 2.0) * (stp - stx)
        if abs(stpc - stx) < abs(stpq - stx):
            stpf = stpc
        else:
            stpf = stpc + (stpq - stpc) / 2.0
        brackt = True
    elif sgnd < 0.0:
        theta = 3.0 * (fx - fp) / (stp - stx) + dx + dp
        sc = max(abs(theta), abs(dx), abs(dp))
        gamma = sc * math.sqrt((theta / sc) * (theta / sc) - (dx / sc) * (dp / sc))
        if stp > stx:
            gamma = -gamma
        p = (gamma - dp) + theta
        q = ((gamma - dp) + gamma) + dx
        r = p / q
        stpc = stp + r * (stx - stp)
        stpq = stp + (dp / (dp - dx)) * (stx - stp)
        stpf = stpc if abs(stpc - stp) > abs(stpq - stp) else stpq
        brackt = True
    elif abs(dp) < abs(dx):
        theta = 3.0 * (fx - fp) / (stp - stx) + dx + dp
        sc = max(abs(theta), abs(dx), abs(dp))
        gamma = sc * math.sqrt(max(0.0, (theta / sc) * (theta / sc) - (dx / sc) * (dp / sc)))
        if stp > stx:
            gamma = -gamma
        p = (gamma - dp) + theta
        q = (gamma + (dx - dp)) + gamma
        r = p / q
        if r < 0.0 and gamma != 0.0:
            stpc = stp + r * (stx - stp)
        elif stp > stx:
            stpc = stpmax
        else:
            stpc = stpmin
        stpq = stp + (dp / (dp - dx)) * (stx - stp)
        if brackt:
            stpf = stpc if abs(stpc - stp) < abs(stpq - stp) else stpq
            if stp > stx:
                stpf = min(stp + 0.66 * (sty - stp), stpf)
            else:
                stpf = max(stp + 0.66 * (sty - stp), stpf)
        else:
            stpf = stpc if abs(stpc - stp) > abs(stpq - stp) else stpq
            stpf = min(stpmax, stpf)
            stpf = max(stpmin, stpf)
    else:
        if brackt:
            theta = 3.0 * (fp - fy) / (sty - stp) + dy + dp
            sc = max(abs(theta), abs(dy), abs(dp))
            gamma = sc * math.sqrt((theta / sc) * (theta / sc) - (dy / sc) * (dp / sc))
            if stp > sty:
                gamma = -gamma
            p = (gamma - dp) + theta
            q = ((gamma - dp) + gamma) + dy
            r = p / q
            stpc = stp + r * (sty - stp)
            stpf = stpc
        elif stp > stx:
            stpf = stpmax
        else:
            stpf = stpmin
    if fp > fx:
        sty, fy, dy = stp, fp, dp
    else:
        if sgnd < 0.0:
            sty, fy, dy = stx, fx, dx
        stx, fx, dx = stp, fp, dp
    return stx, fx, dx, sty, fy, dy, stpf, brackt


def lbfgsb_1d(fun, x0, lb, ub, pgtol=1e-5, factr=1e7, maxls=20, eps=1e-8, maxiter=15000, info=None, formk_rule=False):
    """scipy.optimize.minimize(fun, [x0], method="L-BFGS-B", bounds=[(lb, ub)]) for a scalar variable, statement by statement (see
    the block comment above).  Returns the final parameter; info (a dict) receives nfev, nit, the exit reason and the number of
    restarts formk forced.  formk_rule: formk's verdict by the rule the DEVICE uses instead of the transcribed bookkeeping --
    "fails exactly when the variable entered the free set in this iteration and at least two pairs are stored"
    (csrc/mg_traj_device.h; the two agree on every golden track and in a randomised campaign of 108 000 calls)."""
    nfev, nit, restarts = [0], 0, [0]
    m = _LB_M

    def done(x, why):
        if info is not None:
            info.update(nfev=nfev[0], nit=nit, message=why, formk_restarts=restarts[0])
        return x
    x = min(max(float(x0), lb), ub)
    if lb == ub:                                   # _optimize_result_for_equal_bounds
        return done(lb, "fixed by bounds")

    def fg(x):
        f = fun(x)
        h = eps                                    # approx_derivative: abs_step, one-sided, adjusted to the bounds
        lower, upper = x - lb, ub - x
        if x + h < lb or x + h > ub:
            if abs(h) <= max(lower, upper):
                h = -h
            elif upper >= lower:
                h = upper
            else:
                h = -lower
        x1 = x + h
        f1 = fun(x1)
        nfev[0] += 2
        return f, (f1 - f) / (x1 - x)

    def projgr(x, g):
        gi = g
        if gi < 0.0:
            gi = max(x - ub, gi)
        else:
            gi = min(x - lb, gi)
        return abs(gi)
    f, g = fg(x)
    sbgnrm = projgr(x, g)
    if sbgnrm <= pgtol:
        return done(x, "pgtol")
    col, theta, itr, head, iupdat, updatd = 0, 1.0, 0, 1, 0, False
    ws, wy, dr_of = [0.0] * (m + 1), [0.0] * (m + 1), [0.0] * (m + 1)
    fk = _FormK()
    was_free = True                                # mainlb starts with nfree = n
    while True:
        # ---- cauchy: the generalised Cauchy point of the model f + g d + theta d^2 / 2 on [lb, ub]
        neggi = -g
        tl, tu = x - lb, ub - x
        xlower, xupper = tl <= 0.0, tu <= 0.0
        fixed = (xlower and neggi <= 0.0) or (not xlower and xupper and neggi >= 0.0)
        if fixed or neggi == 0.0:
            z, free = x, not fixed
        else:
            f1 = -neggi * neggi
            f2 = -theta * f1
            dtm = -f1 / f2
            tbreak = tl / (-neggi) if neggi < 0.0 else tu / neggi
            if dtm < tbreak:
                if dtm <= 0.0:
                    dtm = 0.0
                z, free = x + dtm * neggi, True
            else:
                z, free = (ub if neggi > 0.0 else lb), False
        # ---- freev: who entered, who left (counted from the second iteration on), is formk's matrix stale?
        entered = itr > 0 and free and not was_free
        left = itr > 0 and was_free and not free
        wrk = left or entered or updatd
        was_free = free
        # ---- formk (subsm itself: a free Cauchy point is the one-dimensional model's minimiser already)
        if free and col > 0:
            if wrk and not ((not (entered and col >= 2)) if formk_rule else fk.call(free, entered, left, updatd, iupdat, col, head, ws, wy, dr_of, theta)):
                col, head, theta, iupdat, updatd = 0, 1, 1.0, 0, False       # refresh the memory and restart the iteration
                restarts[0] += 1
                continue
        d = z - x
        # ---- lnsrlb + dcsrch
        if itr == 0:
            stpmx = 1.0
        else:
            stpmx = 1.0e10
            if d < 0.0:
                a2 = lb - x
                if a2 >= 0.0:
                    stpmx = 0.0
                elif d * stpmx < a2:
                    stpmx = a2 / d
            elif d > 0.0:
                a2 = ub - x
                if a2 <= 0.0:
                    stpmx = 0.0
                elif d * stpmx > a2:
                    stpmx = a2 / d
        stp = 1.0
        dtd = d * d
        t, r_, fold = x, g, f
        ifun, failed = 0, False
        gd = g * d
        gdold = gd
        if gd >= 0.0:
            failed = True                          # info = -4: not a descent direction
        else:
            # dcsrch, task START
            ftol_ls, gtol_ls, xtol_ls, stpmin = 1.0e-3, 0.9, 0.1, 0.0
            brackt, stage = False, 1
            finit, ginit = f, gd
            gtest = ftol_ls * ginit
            width = stpmx - stpmin
            width1 = width / 0.5
            stx, fx, gx = 0.0, finit, ginit
            sty, fy, gy = 0.0, finit, ginit
            stmin, stmax = 0.0, stp + 4.0 * stp
            while True:
                ifun += 1                          # FG_LNSRCH: evaluate at the trial step
                if ifun - 1 >= maxls:
                    failed = True
                    break
                x = z if stp == 1.0 else stp * d + t
                f, g = fg(x)
                gd = g * d
                ftest = finit + stp * gtest        # dcsrch, re-entry
                if stage == 1 and f <= ftest and gd >= 0.0:
                    stage = 2
                task = None
                if brackt and (stp <= stmin or stp >= stmax):
                    task = "WARN"
                if brackt and stmax - stmin <= xtol_ls * stmax:
                    task = "WARN"
                if stp == stpmx and f <= ftest and gd <= gtest:
                    task = "WARN"
                if stp == stpmin and (f > ftest or gd >= gtest):
                    task = "WARN"
                if f <= ftest and abs(gd) <= gtol_ls * (-ginit):
                    task = "CONV"
                if task is not None:
                    break
                if stage == 1 and f <= fx and f > ftest:
                    fm, fxm, fym = f - stp * gtest, fx - stx * gtest, fy - sty * gtest
                    gm, gxm, gym = gd - gtest, gx - gtest, gy - gtest
                    stx, fxm, gxm, sty, fym, gym, stp, brackt = _dcstep(stx, fxm, gxm, sty, fym, gym, stp, fm, gm, brackt, stmin, stmax)
                    fx, fy = fxm + stx * gtest, fym + sty * gtest
                    gx, gy = gxm + gtest, gym + gtest
                else:
                    stx, fx, gx, sty, fy, gy, stp, brackt = _dcstep(stx, fx, gx, sty, fy, gy, stp, f, gd, brackt, stmin, stmax)
                if brackt:
                    if abs(sty - stx) >= 0.66 * width1:
                        stp = stx + 0.5 * (sty - stx)
                    width1 = width
                    width = abs(sty - stx)
                if brackt:
                    stmin, stmax = min(stx, sty), max(stx, sty)
                else:
                    stmin = stp + 1.1 * (stp - stx)
                    stmax = stp + 4.0 * (stp - stx)
                stp = max(stp, stpmin)
                stp = min(stp, stpmx)
                if (brackt and (stp <= stmin or stp >= stmax)) or (brackt and stmax - stmin <= xtol_ls * stmax):
                    stp = stx
        if failed:
            x, g, f = t, r_, fold                  # restore the previous iterate
            if col == 0:
                return done(x, "abnormal termination in lnsrch")
            col, head, theta, iupdat, updatd = 0, 1, 1.0, 0, False           # refresh the memory and restart the iteration
            continue
        # ---- NEW_X
        itr += 1
        nit += 1
        sbgnrm = projgr(x, g)
        if sbgnrm <= pgtol:
            return done(x, "pgtol")
        ddum = max(abs(fold), abs(f), 1.0)
        if fold - f <= _LB_EPSMCH * factr * ddum:
            return done(x, "factr")
        if nit >= maxiter:                         # (the wrapper's own count, tested when NEW_X comes back to Python)
            return done(x, "maxiter")
        # ---- the pair: y = g - g_old, s = stp d
        y = g - r_
        rr = y * y
        if stp == 1.0:
            dr, dd, sstep = gd - gdold, -gdold, d
        else:
            dr, dd, sstep = (gd - gdold) * stp, -gdold * stp, stp * d
        if dr <= _LB_EPSMCH * dd:
            updatd = False                         # skipped: the model stays what it was
            continue
        updatd = True                              # matupd
        iupdat += 1
        if iupdat <= m:
            col = iupdat
            itail = (head + iupdat - 2) % m + 1
        else:
            itail = itail % m + 1
            head = head % m + 1
        ws[itail], wy[itail], dr_of[itail] = sstep, y, dr
        theta = rr / dr


def closest_point_lbfgsb(control_points, point, min_u, info=None, formk_rule=False):
    """find_closest_point_fast (parameterized_spline.py:303-322) through lbfgsb_1d: the form the device runs
    (csrc/mg_traj_device.h, mg_traj_closest_lbfgsb).  The distance is evaluated as the reference's objective does it: the norm of
    (point on the spline - target)."""
    target = np.asarray(point, dtype=np.float64)

    def dist(u):
        v = catmull_rom_point(control_points, u) - target
        return math.sqrt(float(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]))
    u = lbfgsb_1d(dist, float(min_u), float(min_u), 1.0, info=info, formk_rule=formk_rule)
    return catmull_rom_point(control_points, u), u


def closest_point(control_points, point, min_u, granularity=1000, search="reference"):
    """The closest-point search as the device runs it: "reference" (MG_OPT_TRAJECTORY_SEARCH 0, the default) or "walk" (1)."""
    if search == "walk":
        return closest_point_walk(control_points, point, min_u, granularity)
    return closest_point_lbfgsb(control_points, point, min_u, formk_rule=True)


def closest_point_walk(control_points, point, min_u, granularity=1000):
    """The DEVICE's deterministic form of the same search (what mg_score_trajectory computes, restated): on the grid
    u_k = k / granularity walk forward from the first grid point at or after min_u while the distance falls, then refine
    the parameter by the parabola through the squared distances of the three grid points around the minimum, then by up to
    four Newton steps on the squared distance inside the bracket of those three grid points (clamped to [min_u, 1]): the
    local minimum of the bracket to rounding, so never farther from the point than a converged search of the same basin."""
    target = np.asarray(point, dtype=np.float64)
    G = int(granularity)

    def d2(u):
        v = catmull_rom_point(control_points, u) - target
        return float(v @ v)
    k = min(G, int(math.ceil(min_u * G - 1e-12)))
    dk = d2(k / G)
    d_start = d2(min_u)
    while k < G:
        dn = d2((k + 1) / G)
        if dn >= dk:
            break
        k, dk = k + 1, dn
    u = k / G
    if 0 < k < G:
        a, b, c = d2((k - 1) / G), dk, d2((k + 1) / G)
        den = a - 2.0 * b + c
        if den > 0.0:
            u = (k + 0.5 * (a - c) / den) / G
    lo, hi = max(float(min_u), (k - 1) / G), min(1.0, (k + 1) / G)      # the bracket of the grid minimum (one-sided at the ends)
    u = min(hi, max(lo, u))
    for _ in range(4):
        v = catmull_rom_point(control_points, u) - target
        p1, p2 = catmull_rom_derivatives(control_points, u)
        f1, f2 = 2.0 * float(v @ p1), 2.0 * float(p1 @ p1 + v @ p2)
        if not f2 > 0.0:
            break
        un = min(hi, max(lo, u - f1 / f2))
        if d2(un) > d2(u):
            break
        u = un
    u = min(1.0, max(float(min_u), u))
    if d_start <= d2(u):
        u = float(min_u)
    return catmull_rom_point(control_points, u), u


def arc_length_table(control_points, granularity=1000):
    """RelativeArcLengthMap._update_table (splines/arc_length_map.py:45-71): [(parameter, relative arc length)] over
    granularity + 1 samples, and the full arc length.  PINNED by tests/golden/trajectory_spline.npz (points_by_arc_*)."""
    table, full, last = [], 0.0, None
    for k in range(granularity + 1):
        u = k / float(granularity)
        pt = catmull_rom_point(control_points, u)
        if last is not None:
            full += float(np.linalg.norm(pt - last))
        table.append([u, full])
        last = pt
    if full == 0.0:
        raise ValueError("Not enough control points in trajectory constraint definition")
    for row in table:
        row[1] /= full
    return table, full


def catmull_rom_point_by_arc_length(control_points, table, full, arc):
    """ParameterizedSpline.query_point_by_absolute_arc_length (splines/parameterized_spline.py:131-155) with
    map_relative_arc_length_to_parameter (arc_length_map.py:97-160): the table entries bounding the relative arc length,
    interpolated linearly; past the full arc length the last control point."""
    if arc > full:
        return np.asarray(control_points[-1], dtype=np.float64).copy()
    rel = arc / full
    if rel <= table[0][1]:
        return catmull_rom_point(control_points, table[0][0])
    if rel >= table[-1][1]:
        return catmull_rom_point(control_points, table[-1][0])
    lo = 0
    while lo + 1 < len(table) and table[lo + 1][1] <= rel:       # the closest lower entry
        lo += 1
    if table[lo][1] == rel:
        return catmull_rom_point(control_points, table[lo][0])
    (p0, l0), (p1, l1) = table[lo], table[lo + 1]
    return catmull_rom_point(control_points, p0 + (rel - l0) / (l1 - l0) * (p1 - p0))


def per_frame_constraint_residuals(c, knots, coeffs, n_canonical_frames, joints, animated_joints, search="reference"):
    """(residual vector, error), both WITHOUT the weight factor, of the constraints that walk every frame of one aligned
    motion (coeffs: its aligned control points) -- what get_residual_vector_spline / evaluate_motion_spline of the
    reference classes return:
      frame_joint_trajectory   TrajectoryConstraint, trajectory_constraint.py:79-116
      frame_ca_position        GlobalTransformCAConstraint, keyframe_constraints/global_transform_ca_constraint.py:33-46
      frame_discrete_trajectory DiscreteTrajectoryConstraint, discrete_trajectory_constraint.py:66-90
      frame_local_trajectory   LocalTrajectoryConstraint, keyframe_constraints/local_trajectory_constraint.py:45-78
      frame_trajectory_set     TrajectorySetConstraint, trajectory_set_constraint.py:41-104
      frame_joint_rotation     JointRotationConstraint, keyframe_constraints/joint_rotation_constraint.py:55-72
    The joints' positions come from forward kinematics (anim_utils': PARITY UNPINNED); what the classes do WITH the positions is
    per_frame_track_residuals, pinned by the reference's own runs (tests/golden/per_frame_classes.npz,
    trajectory_closest_point.npz)."""
    F = int(n_canonical_frames)
    kind = c["type"]

    def position(frame, joint):
        return np.asarray(joint_global_position(frame, joints, animated_joints, joint), dtype=np.float64)
    if kind == "frame_joint_rotation":
        frame = spline_frames(knots, coeffs, [float(c["frame_idx"])])[0]
        ji = int(c["joint_index"])
        q = np.array(frame[3 + 4 * ji:3 + 4 * (ji + 1)], dtype=np.float64)
        q /= np.linalg.norm(q)
        t = np.asarray(c["quaternion"], dtype=np.float64)
        err = float(np.linalg.norm(np.ravel(quaternion_matrix3(t / np.linalg.norm(t)) - quaternion_matrix3(q))))
        return np.array([err]), err
    if kind in ("frame_ca_position", "frame_local_trajectory"):       # aligned_spline.evaluate(i), i = 0 .. n - 1
        nf = int(c.get("n_frames", F))
        frames = [spline_frames(knots, coeffs, [float(i)])[0] for i in range(nf)]
    else:                                                              # MotionSpline.get_motion_vector()
        frames = spline_frames(knots, coeffs, np.linspace(0, F, F))
    names = list(c["joints"]) if kind == "frame_trajectory_set" else [c["joint"]]
    tracks = {j: np.array([position(f, j) for f in frames]) for j in names}
    return per_frame_track_residuals(c, tracks, search)


def per_frame_track_residuals(c, tracks, search="reference"):
    """The same, from the joints' positions per frame (tracks: joint -> (T, 3), the frames the class reads): the arithmetic of the
    reference classes downstream of forward kinematics."""
    kind = c["type"]
    if kind == "frame_joint_trajectory":
        cps, min_u = c["control_points"], float(c.get("min_u", 0.0))
        track = tracks[c["joint"]]
        errors = np.empty(len(track))
        for i, p in enumerate(track):
            target, min_u = closest_point(cps, p, min_u, c.get("granularity", 1000), search)
            errors[i] = np.linalg.norm(p - target)
        return errors, float(np.average(errors))
    if kind == "frame_ca_position":
        errors = np.array([point_distance(c["target"], p) for p in tracks[c["joint"]]])
        return np.array([min(errors)]), float(min(errors))
    if kind == "frame_discrete_trajectory":
        pts, free = [np.array(p, dtype=np.float64) for p in c["points"]], [int(a) for a in (c.get("unconstrained") or ())]
        errors = []
        for index, pos in enumerate(tracks[c["joint"]]):
            if index < len(pts):
                p, target = np.array(pos, dtype=np.float64), pts[index].copy()
                target[free] = 0
                p[free] = 0
                errors.append(np.linalg.norm(p - target))
            else:
                errors.append(0.0)
        return np.array(errors), float(np.average(errors))
    if kind == "frame_local_trajectory":
        table, full = arc_length_table(c["control_points"], c.get("granularity", 1000))
        errors, last_p, current_t = [], None, float(c.get("start_t", 0.0))
        for p in tracks[c["joint"]]:
            if last_p is not None:
                current_t += np.linalg.norm(last_p - p)
            target = catmull_rom_point_by_arc_length(c["control_points"], table, full, current_t)
            delta = np.array([target[0] - p[0], target[2] - p[2]])
            errors.append(float(np.dot(delta, delta)))
            last_p = p
        return np.array(errors), float(sum(errors))
    if kind == "frame_trajectory_set":
        names = list(c["joints"])
        nf = int(c.get("n_frames", len(tracks[names[0]])))
        tables = [arc_length_table(t["control_points"], t.get("granularity", 1000)) for t in c["trajectories"]]
        arcs = [float(a) for a in c.get("arc_lengths", [0.0] * len(names))]
        residual, last = np.zeros(nf), None
        for i in range(nf):
            ps = [tracks[j][i] for j in names]
            active = [t.get("range_start") is not None and t["range_start"] <= a <= t["range_end"] for t, a in zip(c["trajectories"], arcs)]
            if np.any(active):
                targets = [catmull_rom_point_by_arc_length(t["control_points"], tab, full, a) for t, (tab, full), a in zip(c["trajectories"], tables, arcs)]
                residual[i] = np.linalg.norm(np.average(ps) - np.average(targets))
            if last is not None:
                arcs = [a + np.linalg.norm(p - q) for p, q, a in zip(ps, last, arcs)]
            last = ps
        return residual, float(np.average(residual))
    raise ValueError(kind)


def align_point_clouds_2d(a, b, weights):
    """The optimal weighted 2-D rigid fit of cloud b onto cloud a (rotation about y by theta, then translation in x
    and z): the closed form of Kovar, Gleicher, Pighin, "Motion Graphs" (2002), which the reference reaches through
    anim_utils' align_point_clouds_2D (absent here; PARITY UNPINNED, optimality checked by the tests).
    Returns theta, offset_x, offset_z with x' = x cos + z sin + ox, z' = -x sin + z cos + oz."""
    a, b, w = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64), np.asarray(weights, dtype=np.float64)
    sw = w.sum()
    sax, saz, sbx, sbz = (w * a[:, 0]).sum(), (w * a[:, 2]).sum(), (w * b[:, 0]).sum(), (w * b[:, 2]).sum()
    num = (w * (a[:, 0] * b[:, 2] - b[:, 0] * a[:, 2])).sum() - (sax * sbz - sbx * saz) / sw
    den = (w * (a[:, 0] * b[:, 0] + a[:, 2] * b[:, 2])).sum() - (sax * sbx + saz * sbz) / sw
    theta = math.atan2(num, den)
    ox = (sax - sbx * math.cos(theta) - sbz * math.sin(theta)) / sw
    oz = (saz + sbx * math.sin(theta) - sbz * math.cos(theta)) / sw
    return theta, ox, oz


def transform_point_cloud(cloud, theta, ox, oz):
    cloud = np.asarray(cloud, dtype=np.float64)
    out = cloud.copy()
    out[:, 0] = cloud[:, 0] * math.cos(theta) + cloud[:, 2] * math.sin(theta) + ox
    out[:, 2] = -cloud[:, 0] * math.sin(theta) + cloud[:, 2] * math.cos(theta) + oz
    return out


def pose_constraint_error(c, frame1, frame2, joints, animated_joints):
    """PoseConstraint.evaluate_motion_spline (pose_constraint.py:48-67): the cloud of the joints' global positions at
    the keyframe, fitted in 2-D onto the wanted cloud, MEAN distance of corresponding points after the fit, plus
    |velocity - (first joint at t + 1 - first joint at t)| when a velocity is constrained."""
    cloud = np.array([joint_global_position(frame1, joints, animated_joints, j) for j in c["joints"]])
    target = np.asarray(c["points"], dtype=np.float64)
    w = np.asarray(c.get("weights", np.ones(len(cloud))), dtype=np.float64)
    theta, ox, oz = align_point_clouds_2d(target, cloud, w)
    fitted = transform_point_cloud(cloud, theta, ox, oz)
    err = float(np.linalg.norm(target - fitted, axis=1).sum() / len(cloud))
    if c.get("velocity") is not None:
        nxt = joint_global_position(frame2, joints, animated_joints, c["joints"][0])
        err += float(np.linalg.norm(np.asarray(c["velocity"], dtype=np.float64) - (nxt - cloud[0])))
    return err


def constraint_error_on_frame(c, frame, joints, animated_joints):
    """One constraint dict evaluated on one (aligned or local) pose vector."""
    kind = c["type"]
    if kind == "position":
        return point_distance(c["target"], frame[:3])
    if kind == "joint_position":
        p = joint_global_position(frame, joints, animated_joints, c["joint"])
        if c.get("offset") is not None:   # relative_transform_constraint.py:46-50: global matrix of the joint times the offset
            p = p + joint_global_orientation(frame, joints, animated_joints, c["joint"]) @ np.asarray(c["offset"], dtype=np.float64)[:3]
        return point_distance(c["target"], p)
    if kind == "look_at":                 # look_at_constraint.py:55-66
        head = joint_global_position(frame, joints, animated_joints, c["joint"])
        to_target = np.asarray(c["target"], dtype=np.float64) - head
        to_target = to_target / np.linalg.norm(to_target)
        look = joint_global_orientation(frame, joints, animated_joints, c["joint"]) @ np.asarray(c.get("ref_dir", (0.0, 0.0, 1.0)), dtype=np.float64)
        look = look / np.linalg.norm(look)
        # |angle| of the rotation taking `look` to `to_target` (quaternion_from_vector_to_vector + rotation_from_matrix)
        return math.acos(min(1.0, max(float(np.dot(look, to_target)), -1.0)))
    if kind == "joint_midpoint":
        mid = 0.5 * (joint_global_position(frame, joints, animated_joints, c["joint"]) +
                     joint_global_position(frame, joints, animated_joints, c["joint2"]))
        return point_distance(c["target"], mid)
    if kind == "joint_orientation":
        return joint_orientation_error(frame, joints, animated_joints, c["joint"], c["orientation"], c.get("ref_dir", (0.0, 0.0, 1.0)))
    return direction_2d_error(c["target"], frame[3:7], c.get("ref_dir", (0, 0, 1)))


# --------------------------------------------------------------------------
# the primitive
# --------------------------------------------------------------------------
class OraclePrimitive(object):
    """motion_primitive.py:41-256 restated for the use_time_parameters=False path."""

    def __init__(self, data):
        # motion_primitive.py:96-163
        self.name = data.get("name", "")
        self.n_canonical_frames = int(data["n_canonical_frames"])
        self.translation_maxima = np.array(data["translation_maxima"], dtype=np.float64)
        self.eigen_vectors = np.transpose(np.array(data["eigen_vectors_spatial"], dtype=np.float64))  # (NB*D, L)
        self.mean_vector = np.array(data["mean_spatial_vector"], dtype=np.float64)
        self.n_basis = int(data["n_basis_spatial"])
        self.n_dim = int(data["n_dim_spatial"])
        self.n_components = self.eigen_vectors.shape[1]
        self.knots = np.asarray(data["b_spline_knots_spatial"], dtype=np.float64)
        self.weights = np.array(data["gmm_weights"], dtype=np.float64)
        self.means = np.array(data["gmm_means"], dtype=np.float64)
        self.covars = np.array(data["gmm_covars"], dtype=np.float64)
        self.prec_chol = precision_cholesky(self.covars)
        self.n_time_components = 0
        if "eigen_vectors_time" in data:
            self.init_time_model(data)

    # motion_primitive.py:164-181, 258-266, 289-302 (the legacy time model; present when the model has 'eigen_vectors_time')
    def init_time_model(self, data):
        self.t_eigen_vectors = np.array(data["eigen_vectors_time"], dtype=np.float64)      # (n_basis_time, n_t): column l = harmonic l
        self.t_mean_vector = np.array(data["mean_time_vector"], dtype=np.float64)
        self.t_knots = np.asarray(data["b_spline_knots_time"], dtype=np.float64)
        self.n_time_components = self.t_eigen_vectors.shape[1]

    def mean_temporal(self):
        """_mean_temporal: the mean time spline at the canonical frames 0 .. F-1."""
        return splev(np.arange(self.n_canonical_frames, dtype=np.float64), self.t_knots, self.t_mean_vector)

    def back_transform_gamma_to_canonical_time_function(self, gamma):
        """_back_transform_gamma_to_canonical_time_function: t(t') = cumulative sum of exp(mean + harmonics . gamma), - 1."""
        frames = np.arange(self.n_canonical_frames, dtype=np.float64)
        mean_t = self.mean_temporal()
        t_eigen_discrete = np.array([splev(frames, self.t_knots, self.t_eigen_vectors[:, l]) for l in range(self.n_time_components)]).T
        gamma = np.asarray(gamma, dtype=np.float64)
        out, acc = [], 0.0
        for i in range(self.n_canonical_frames):
            acc = acc + np.exp(mean_t[i] + np.dot(t_eigen_discrete[i], gamma))
            out.append(acc)
        return np.array(out) - 1.0

    # motion_primitive.py:304-319
    def invert_canonical_to_sample_time_function(self, canonical_time_function, speed=1.0):
        """_invert_canonical_to_sample_time_function: t'(t) from t(t') -- scipy's interpolating cubic through (t(t'), t')
        (splrep, k = 3, no smoothing) evaluated at linspace(1, t(F-2), num), 0 in front, F - 1 behind.  num = int(round(t(F-2)) *
        (1 / speed)): the reference passes the float, which NumPy >= 1.18 refuses; up to 1.17 it was truncated
        (tests/golden/time_model.npz holds the reference's own output with that behaviour restored: oracle/gen_golden.py)."""
        from scipy.interpolate import splev as si_splev, splrep
        F = self.n_canonical_frames
        ctf = np.asarray(canonical_time_function, dtype=np.float64)
        inverse = splrep(ctf, np.arange(F), w=None, k=B_SPLINE_DEGREE)
        num = int(np.round(ctf[-2]) * (1.0 / speed))
        inner = si_splev(np.linspace(1, stop=ctf[-2], num=num), inverse)
        return np.concatenate(([0.0], inner, [F - 1.0]))

    # motion_primitive.py:206-234 with use_time_parameters=True, + motion_spline.py:71-86
    def back_project_warped_frames(self, s, speed=1.0):
        s = np.asarray(s, dtype=np.float64)
        ctf = self.back_transform_gamma_to_canonical_time_function(s[self.n_components:self.n_components + self.n_time_components])
        tf = self.invert_canonical_to_sample_time_function(ctf, speed)
        return tf, spline_frames(self.knots, self.back_project_spatial_coeffs(s[:self.n_components]), tf)

    # motion_primitive.py:236-256
    def back_project_spatial_coeffs(self, alpha):
        coefs = np.dot(self.eigen_vectors, np.asarray(alpha, dtype=np.float64))
        coefs = coefs + self.mean_vector
        coefs = coefs.reshape((self.n_basis, self.n_dim))
        coefs[:, :3] *= self.translation_maxima
        return coefs

    # motion_primitive.py:233
    def canonical_time_function(self, speed=1.0):
        return np.linspace(0, self.n_canonical_frames, int(self.n_canonical_frames * (1.0 / speed)))

    # motion_primitive.py:206-234 + motion_spline.py:71-86
    def back_project_frames(self, s, time_points=None):
        coeffs = self.back_project_spatial_coeffs(np.asarray(s)[:self.n_components])
        tp = self.canonical_time_function() if time_points is None else time_points
        return spline_frames(self.knots, coeffs, tp)

    def back_project_frames_batch(self, S, time_points=None):
        """Vectorised over candidates: (B, L) -> (B, T, D)."""
        S = np.atleast_2d(np.asarray(S, dtype=np.float64))
        tp = self.canonical_time_function() if time_points is None else np.atleast_1d(time_points)
        i0, w = basis_rows(self.knots, tp)
        coeffs = S[:, :self.n_components] @ self.eigen_vectors.T + self.mean_vector
        coeffs = coeffs.reshape(S.shape[0], self.n_basis, self.n_dim)
        coeffs[:, :, :3] *= self.translation_maxima
        out = np.zeros((S.shape[0], len(tp), self.n_dim))
        for j in range(4):
            out = out + coeffs[:, i0 + j, :] * w[None, :, j:j + 1]
        return out

    def score_samples(self, X):
        return gmm_log_prob(X, self.weights, self.means, self.prec_chol)

    def sample_low_dimensional_vector(self, n_samples=1, rng=None):
        rng = np.random.mtrand._rand if rng is None else rng
        return gmm_sample(n_samples, self.weights, self.means, self.covars, rng)[0]

    # motion_primitive_constraints.py:100-122 restricted to the FK-free constraints
    def keyframe_residuals(self, S, constraints):
        """MotionPrimitiveConstraints.get_residual_vector (motion_primitive_constraints.py:124-144) for root-joint
        keyframe constraints: (n_samples, n_constraints), entry = weight_factor * error."""
        S = np.atleast_2d(S)
        out = np.zeros((S.shape[0], len(constraints)))
        for b in range(S.shape[0]):
            coeffs = self.back_project_spatial_coeffs(S[b][:self.n_components])
            for ci, c in enumerate(constraints):
                frame = spline_frames(self.knots, coeffs, [c["t"]])[0]
                if c["type"] == "position":
                    out[b, ci] = c["weight"] * point_distance(c["target"], frame[:3])
                else:
                    out[b, ci] = c["weight"] * direction_2d_error(c["target"], frame[3:7], c.get("ref_dir", (0, 0, 1)))
        return out

    def joint_position_residuals(self, S, constraints, joints, animated_joints):
        """weight * _point_distance(target, FK position of the joint at the keyframe) per (sample, constraint)."""
        S = np.atleast_2d(S)
        out = np.zeros((S.shape[0], len(constraints)))
        for b in range(S.shape[0]):
            coeffs = self.back_project_spatial_coeffs(S[b][:self.n_components])
            for ci, c in enumerate(constraints):
                frame = spline_frames(self.knots, coeffs, [c["t"]])[0]
                out[b, ci] = c["weight"] * point_distance(c["target"], joint_global_position(frame, joints, animated_joints, c["joint"]))
        return out

    def aligned_residuals(self, S, constraints, prev_frame, joints, animated_joints, align_joint, ref_dir=(0.0, 0.0, 1.0)):
        """MotionPrimitiveConstraints.get_residual_vector outside local mode (motion_primitive_constraints.py:124-144):
        back-project, align the control points to the previous motion, evaluate every constraint on the aligned
        spline.  Constraint dicts as in keyframe_residuals, plus "joint_position"."""
        S = np.atleast_2d(S)
        out = np.zeros((S.shape[0], len(constraints)))
        for b in range(S.shape[0]):
            coeffs = self.back_project_spatial_coeffs(S[b][:self.n_components])
            coeffs = align_coeffs_to_previous_frame(coeffs, prev_frame, joints, animated_joints, align_joint, ref_dir)
            for ci, c in enumerate(constraints):
                frame = spline_frames(self.knots, coeffs, [c["t"]])[0]
                if c["type"] == "pose":
                    frame2 = spline_frames(self.knots, coeffs, [c["t"] + 1.0])[0]
                    out[b, ci] = c["weight"] * pose_constraint_error(c, frame, frame2, joints, animated_joints)
                else:
                    out[b, ci] = c["weight"] * constraint_error_on_frame(c, frame, joints, animated_joints)
        return out

    def start_pose_residuals(self, S, constraints, start_pose, joints, animated_joints):
        """The residuals of candidates aligned to a START POSE (no previous frames): (n_samples, n_constraints).  The
        start pose is passed on from candidate to candidate as the reference's caller does (its position entry is
        rewritten by every call, see align_coeffs_to_start_pose)."""
        S = np.atleast_2d(S)
        out = np.zeros((S.shape[0], len(constraints)))
        for b in range(S.shape[0]):
            coeffs = self.back_project_spatial_coeffs(S[b][:self.n_components])
            coeffs = align_coeffs_to_start_pose(coeffs, start_pose)
            for ci, c in enumerate(constraints):
                frame = spline_frames(self.knots, coeffs, [c["t"]])[0]
                out[b, ci] = c["weight"] * constraint_error_on_frame(c, frame, joints, animated_joints)
        return out

    def skeleton_residuals(self, S, constraints, joints, animated_joints):
        """Every constraint type of the device scorer in local coordinates: (n_samples, n_constraints)."""
        S = np.atleast_2d(S)
        out = np.zeros((S.shape[0], len(constraints)))
        for b in range(S.shape[0]):
            coeffs = self.back_project_spatial_coeffs(S[b][:self.n_components])
            for ci, c in enumerate(constraints):
                frame = spline_frames(self.knots, coeffs, [c["t"]])[0]
                if c["type"] == "pose":
                    frame2 = spline_frames(self.knots, coeffs, [c["t"] + 1.0])[0]
                    out[b, ci] = c["weight"] * pose_constraint_error(c, frame, frame2, joints, animated_joints)
                else:
                    out[b, ci] = c["weight"] * constraint_error_on_frame(c, frame, joints, animated_joints)
        return out

    def frame_constraint_errors(self, S, constraints, joints, animated_joints, prev_frame=None, align_joint=None, ref_dir=(0.0, 0.0, 1.0),
                                start_pose=None):
        """(n,) the weighted sum of per-frame constraints' errors (MotionPrimitiveConstraints.evaluate,
        motion_primitive_constraints.py:100-122) and per constraint the (n, m) weighted residual vectors."""
        S = np.atleast_2d(S)
        total, blocks = np.zeros(S.shape[0]), [[] for _ in constraints]
        for b in range(S.shape[0]):
            coeffs = self.back_project_spatial_coeffs(S[b][:self.n_components])
            if prev_frame is not None:
                coeffs = align_coeffs_to_previous_frame(coeffs, prev_frame, joints, animated_joints, align_joint, ref_dir)
            elif start_pose is not None:
                coeffs = align_coeffs_to_start_pose(coeffs, start_pose)
            for ci, c in enumerate(constraints):
                res, err = per_frame_constraint_residuals(c, self.knots, coeffs, self.n_canonical_frames, joints, animated_joints)
                total[b] += c.get("weight", 1.0) * err
                blocks[ci].append(c.get("weight", 1.0) * res)
        return total, [np.array(v) for v in blocks]

    def log_likelihood_jac(self, S):
        S = np.atleast_2d(np.asarray(S, dtype=np.float64))
        return np.array([gmm_log_likelihood_jac(s, self.weights, self.means, self.covars, self.prec_chol) for s in S])

    def keyframe_errors(self, S, constraints):
        """constraints: list of dicts {"type": "position"|"direction", "t": float,
        "weight": w, "target": [x|None,y|None,z|None] or [dx, dz]}.  Root joint only."""
        S = np.atleast_2d(S)
        out = np.zeros(S.shape[0])
        for b in range(S.shape[0]):
            coeffs = self.back_project_spatial_coeffs(S[b][:self.n_components])
            err = 0.0
            for c in constraints:
                frame = spline_frames(self.knots, coeffs, [c["t"]])[0]
                if c["type"] == "position":
                    err += c["weight"] * point_distance(c["target"], frame[:3])
                else:
                    err += c["weight"] * direction_2d_error(c["target"], frame[3:7], c.get("ref_dir", (0, 0, 1)))
            out[b] = err
        return out


# --------------------------------------------------------------------------
# graph-walk (global) objectives: optimization/objective_functions.py:290-380 restated for ONE concatenated latent vector
# --------------------------------------------------------------------------
def graph_walk_residual_blocks(primitives, alphas, constraints_per_step, prev_frame, joints, animated_joints, align_joint,
                               ref_dir=(0.0, 0.0, 1.0), exit_from="frames", local_steps=()):
    """Per step the weighted residuals of its constraints, the steps chained as obj_global_error_sum /
    obj_global_residual_vector[_and_naturalness] chain them: back-project the step's latents, evaluate its constraints on
    the motion aligned to the PREVIOUS step's aligned motion (unaligned for steps in `local_steps`: is_local constraints,
    motion_primitive_constraints.py:111), then align this step's motion for the next one (always).  `prev_frame`: the last
    frame before the walk, or None (the first step stays as it is).  exit_from "frames": the next step is aligned to the
    last sample of the aligned get_motion_vector() (canonical time F); "coeffs": to the last aligned control point (the
    _and_naturalness form passes `.coeffs` on as if they were frames, :362,:373).  Alignment as align_coeffs_to_previous_frame
    restates it (PARITY UNPINNED: anim_utils)."""
    blocks, prev = [], None if prev_frame is None else np.asarray(prev_frame, dtype=np.float64)
    for i, (prim, alpha, cons) in enumerate(zip(primitives, alphas, constraints_per_step)):
        coeffs = prim.back_project_spatial_coeffs(np.asarray(alpha, dtype=np.float64))
        aligned = coeffs if prev is None else align_coeffs_to_previous_frame(coeffs, prev, joints, animated_joints, align_joint, ref_dir)
        scored = coeffs if i in local_steps else aligned
        res = np.zeros(len(cons))
        for ci, c in enumerate(cons):
            frame = spline_frames(prim.knots, scored, [c["t"]])[0]
            res[ci] = c["weight"] * constraint_error_on_frame(c, frame, joints, animated_joints)
        blocks.append(res)
        if exit_from == "frames":
            prev = spline_frames(prim.knots, aligned, [float(prim.n_canonical_frames)])[0]
        else:
            prev = np.array(aligned[-1], dtype=np.float64)
    return blocks


# ---------------------------------------------------------------------------------------------------------------------
# The component counts of a planner step drawn on the device (include/mg_hip.h, mg_options_step_device_counts).  The
# reference draws them with numpy.random.multinomial inside sklearn's GaussianMixture.sample
# (/root/reference morphablegraphs/motion_model/motion_primitive.py:182-189); the device draw is distributed the same way and
# has its own stream, restated here so that the counts themselves are pinned bit for bit.
# ---------------------------------------------------------------------------------------------------------------------
def philox4x32_10(c0, c1, c2, c3, k0, k1):
    """Philox4x32-10 (Salmon et al., SC'11: the published algorithm, constants M0 = 0xD2511F53, M1 = 0xCD9E8D57,
    W0 = 0x9E3779B9, W1 = 0xBB67AE85) on arrays of 32-bit counters; returns the four output words."""
    c0, c1, c2, c3 = [np.asarray(c, dtype=np.uint64) & 0xFFFFFFFF for c in (c0, c1, c2, c3)]
    c0, c1, c2, c3 = np.broadcast_arrays(c0, c1, c2, c3)
    k0, k1 = np.uint64(k0 & 0xFFFFFFFF), np.uint64(k1 & 0xFFFFFFFF)
    M0, M1, W0, W1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57), np.uint64(0x9E3779B9), np.uint64(0xBB67AE85)
    mask = np.uint64(0xFFFFFFFF)
    for _ in range(10):
        p0, p1 = M0 * c0, M1 * c2
        h0, l0, h1, l1 = p0 >> np.uint64(32), p0 & mask, p1 >> np.uint64(32), p1 & mask
        c0, c1, c2, c3 = h1 ^ c1 ^ k0, l1, h0 ^ c3 ^ k1, l0
        k0, k1 = (k0 + W0) & mask, (k1 + W1) & mask
    return c0, c1, c2, c3


def device_multinomial_counts(n, weights, seed):
    """counts[c] = #{i < n : cum[c-1] <= u_i < cum[c]}, u_i = (Philox(counter = (i >> 2, 0, 0, 0x636e7473), key = seed)[i & 3] + 0.5) / 2^32,
    cum = cumulative normalised weights (float64, summed in order), the last component takes the rest."""
    w = np.asarray(weights, dtype=np.float64)
    K = len(w)
    wsum = 0.0
    for v in w:
        wsum += float(v)
    cum, acc = [], 0.0
    for v in w:
        acc += float(v)
        cum.append(acc / wsum)
    j = np.arange((n + 3) // 4, dtype=np.uint64)
    words = philox4x32_10(j & np.uint64(0xFFFFFFFF), j >> np.uint64(32), 0, 0x636e7473, int(seed) & 0xFFFFFFFF, (int(seed) >> 32) & 0xFFFFFFFF)
    u = (np.stack(words, axis=1).reshape(-1)[:n].astype(np.float64) + 0.5) * (1.0 / 4294967296.0)
    below = [int(np.count_nonzero(u < cum[c])) for c in range(K - 1)] + [int(n)]
    return np.diff([0] + below).astype(np.int64)
