"""Batched objective functions for the numerical optimizers: the GPU forms of
reference morphablegraphs/motion_generator/optimization/objective_functions.py for root-joint keyframe
constraints, in local coordinates or aligned to the previous motion on the device.

Every function keeps the reference's name, `data` tuple layout and return scaling, but takes a whole batch of
latent vectors (n, L) instead of one `s` and returns one row / value per sample, so that a finite-difference
Jacobian (scipy `leastsq` / `approx_fprime` evaluate the objective L+1 times per iteration,
reference least_squares.py:35-64) or a population of starting points is ONE launch.  A 1-D `s` is accepted and
gives the reference's shapes back.

`data[0]` is the motion primitive (anything exposing the HIP primitive as `._prim` or `.motion_primitive._prim`),
`data[1]` the constraint list (or an object with `.constraints`), `data[2]` the previous frames or None: with
previous frames and constraints that are not `is_local` every sample is aligned to the previous motion on the device
(candidate_scoring.alignment_from_prev_frames).  There is no CPU fallback.
"""
import numpy as np

from . import _capi
from .candidate_scoring import (constraints_to_device_form, cached_constraint_set, alignment_from_prev_frames, group_residuals,
                                split_trajectories, cached_trajectory)


def _prim_of(motion_primitive):
    if hasattr(motion_primitive, "_prim"):
        return motion_primitive._prim
    if hasattr(motion_primitive, "motion_primitive") and hasattr(motion_primitive.motion_primitive, "_prim"):
        return motion_primitive.motion_primitive._prim
    if isinstance(motion_primitive, _capi.Primitive):
        return motion_primitive
    raise TypeError("objective_functions: %r does not wrap a HIP primitive" % (type(motion_primitive).__name__,))


def _constraint_list(mp_constraints):
    return mp_constraints.constraints if hasattr(mp_constraints, "constraints") else mp_constraints


def _batch(s):
    s = np.asarray(s)
    return (s[None, :], True) if s.ndim == 1 else (s, False)


def _note(mp_constraints, min_error, n):
    if hasattr(mp_constraints, "min_error"):
        mp_constraints.min_error = min_error
    if hasattr(mp_constraints, "evaluations"):
        mp_constraints.evaluations += n


def _residuals(prim, mp_constraints, S, prev_frames=None, sums=False):
    """(n, n_columns) weighted residuals like MotionPrimitiveConstraints.get_residual_vector
    (motion_primitive_constraints.py:124-144): one column per keyframe residual, then -- the optimizers are invariant to
    the order -- one column per canonical frame for every trajectory constraint (trajectory_constraint.py:91-115).
    sums=True: (n,) what MotionPrimitiveConstraints.evaluate adds up (:100-122) -- a keyframe constraint's weighted
    error, a trajectory constraint's weighted AVERAGE distance (:79-86)."""
    clist = constraints_to_device_form(_constraint_list(mp_constraints))
    if len(clist) == 0:
        return np.zeros(len(S)) if sums else np.zeros((len(S), 0))
    skeleton = getattr(mp_constraints, "hip_skeleton", None)
    alignment = alignment_from_prev_frames(prev_frames, mp_constraints, skeleton)
    from .frame_constraints import split_frame_constraints, frame_constraints_errors
    clist, frame_list = split_frame_constraints(clist)
    keyframes, trajectories = split_trajectories(clist)
    if trajectories and alignment is not None and alignment.get("joint", 0) not in (0, _capi.MG_ALIGN_START_POSE):
        raise NotImplementedError("trajectory constraints in global coordinates need the root joint as aligning node")
    blocks, total = [], np.zeros(len(S))
    if keyframes:
        cset = cached_constraint_set(prim, keyframes, skeleton, alignment)
        res = group_residuals(keyframes, prim.score_constraint_residuals(cset, S))
        blocks.append(res)
        total = total + res.sum(axis=1)
    for c in trajectories:
        err, res = prim.score_trajectory(cached_trajectory(prim, c), S, c.get("min_u", 0.0), c.get("weight", 1.0), alignment, residuals=True)
        blocks.append(res)
        total = total + err
    if frame_list:   # per-frame constraints (other joints' trajectories, collision avoidance, ...): tracks from the device, arithmetic on the host
        err, fblocks = frame_constraints_errors(prim, S, frame_list, skeleton, alignment)
        blocks += fblocks
        total = total + err
    if sums:
        return total
    return np.hstack(blocks) if blocks else np.zeros((len(S), 0))


def obj_spatial_error_sum(s, data):
    """objective_functions.py:141-159: MotionPrimitiveConstraints.evaluate per sample -> (n,) (float for 1-D s)."""
    motion_primitive, mp_constraints, prev_frames = data[:3]
    S, single = _batch(s)
    err = _residuals(_prim_of(motion_primitive), mp_constraints, S, prev_frames, sums=True)
    _note(mp_constraints, float(err[-1]) if len(err) else 0.0, len(S))
    return float(err[0]) if single else err


def log_likelihood_jac(s, gmm_or_primitive):
    """objective_functions.py:95-107: sum_k N_k(s) w_k Sigma_k^-1 (s - mu_k) / p(s) per row (= -grad log p)."""
    S, single = _batch(s)
    jac = _prim_of(gmm_or_primitive).gmm_log_prob_jac(S)
    return jac[0] if single else jac


def _objective_in_one_launch(prim, mp_constraints, S, prev_frames, error_scale, quality_scale):
    """(objective, constraint errors) from mg_objective_error_and_naturalness -- the mixture kernel scores the keyframe constraints
    on the latent tile it holds: one launch, the latents read once -- for the sets that kernel carries (root position / 2-D
    direction constraints, local or aligned through the root joint); None for everything else (the caller then makes the two
    calls, which give the same numbers)."""
    clist = constraints_to_device_form(_constraint_list(mp_constraints))
    if len(clist) == 0 or any(c.get("type") not in ("position", "direction") for c in clist):
        return None
    skeleton = getattr(mp_constraints, "hip_skeleton", None)
    alignment = alignment_from_prev_frames(prev_frames, mp_constraints, skeleton)
    cset = cached_constraint_set(prim, clist, skeleton, alignment)
    try:
        obj, err, _ = prim.objective(cset, np.asarray(S, dtype=np.float64), error_scale, quality_scale)
    except _capi.MGError as e:
        if e.status != -4:        # MG_ERR_UNSUPPORTED: a shape the one-launch kernel does not carry
            raise
        return None
    return obj, err


def obj_spatial_error_sum_and_naturalness(s, data):
    """objective_functions.py:162-184: error_scale * spatial_error + quality_scale * (-log p(s)).
    (The reference function computes this value and then falls off its end without `return`, so scipy receives
    None there; the batched form returns the value it computes.)"""
    motion_primitive, mp_constraints, prev_frames, error_scale, quality_scale = data[0], data[1], data[2], data[-3], data[-2]
    S, single = _batch(s)
    prim = _prim_of(motion_primitive)
    fused = _objective_in_one_launch(prim, mp_constraints, S, prev_frames, error_scale, quality_scale)
    if fused is not None:
        err, spatial = fused
    else:
        spatial = _residuals(prim, mp_constraints, S, prev_frames, sums=True)
        err = error_scale * spatial + (-prim.gmm_log_prob(S.astype(np.float64))) * quality_scale
    _note(mp_constraints, float(spatial[-1]) if len(spatial) else 0.0, len(S))
    return float(err[0]) if single else err


def spatial_error_jac(s, data, epsilon=1e-7):
    """The kinematic part of obj_spatial_error_sum_and_naturalness_jac (objective_functions.py:207):
    scipy approx_fprime's forward differences of obj_spatial_error_sum, (f(s + eps e_i) - f(s)) / eps, with all
    n * (L + 1) evaluations in one launch -> (n, L)."""
    motion_primitive, mp_constraints, prev_frames = data[:3]
    S, single = _batch(s)
    S = np.asarray(S, dtype=np.float64)
    n, L = S.shape
    pert = np.repeat(S[:, None, :], L + 1, axis=1)           # (n, L+1, L): row 0 unperturbed
    pert[:, np.arange(1, L + 1), np.arange(L)] += epsilon
    f = _residuals(_prim_of(motion_primitive), mp_constraints, pert.reshape(n * (L + 1), L), prev_frames, sums=True).reshape(n, L + 1)
    if hasattr(mp_constraints, "evaluations"):
        mp_constraints.evaluations += n * (L + 1)
    jac = (f[:, 1:] - f[:, :1]) / epsilon
    return jac[0] if single else jac


def obj_spatial_error_sum_and_naturalness_jac(s, data, epsilon=1e-7):
    """objective_functions.py:187-208: logLikelihood_jac * quality_scale + kinematic_jac * error_scale, the first
    analytic (mixture), the second by forward differences.  NB the reference reads error_scale = data[-1] and
    quality_scale = data[-2] here (not [-3], [-2] as in the objective); kept."""
    error_scale, quality_scale = data[-1], data[-2]
    S, single = _batch(s)
    jac = log_likelihood_jac(S, data[0]) * quality_scale + spatial_error_jac(S, data, epsilon) * error_scale
    return jac[0] if single else jac


def _pad(res, n_variables):
    if res.shape[1] < n_variables:   # `while n_error_values < n_variables: residual_vector.append(0)`
        res = np.hstack([res, np.zeros((res.shape[0], n_variables - res.shape[1]))])
    return res


def obj_spatial_error_residual_vector(s, data):
    """objective_functions.py:209-236: weighted residual of every constraint, zero-padded to n_variables columns,
    divided by init_error_sum -> (n, max(n_constraints, L))."""
    motion_primitive, mp_constraints, prev_frames, error_scale, quality_scale, init_error_sum = data
    S, single = _batch(s)
    res = _residuals(_prim_of(motion_primitive), mp_constraints, S, prev_frames)
    _note(mp_constraints, float(res[-1].sum()) if len(res) else 0.0, len(S))
    out = _pad(res, S.shape[1]) / init_error_sum
    return out[0] if single else out


def obj_spatial_error_residual_vector_and_naturalness(s, data):
    """objective_functions.py:239-267: (residual_i * error_scale - log p(s) * quality_scale), zero-padded to
    n_variables columns, divided by init_error_sum."""
    mp, mp_constraints, prev_frames, error_scale, quality_scale, init_error_sum = data
    S, single = _batch(s)
    prim = _prim_of(mp)
    nll = -prim.gmm_log_prob(S.astype(np.float64)) * quality_scale
    res = _residuals(prim, mp_constraints, S, prev_frames)
    _note(mp_constraints, float(res[-1].sum()) if len(res) else 0.0, len(S))
    out = _pad(res * error_scale + nll[:, None], S.shape[1]) / init_error_sum
    return out[0] if single else out


# ---------------------------------------------------------------------------------------------------------------------
# Graph-walk (global) objectives: reference optimization/objective_functions.py:290-380, driven by
# motion_generator/graph_walk_optimizer.py:78-105.  `s` concatenates the spatial latents of the walk's steps; per step the
# reference back-projects, evaluates the step's constraints against the PREVIOUS step's aligned frames and aligns this step's
# motion for the next one -- a chain.  Batched: S (n, sum L_i); a candidate's chain is its own (its step i is aligned to ITS
# step i - 1), so per step ONE launch scores all n candidates with a per-candidate alignment record
# (mg_score_constraint_residuals_chained) and returns, beside the residuals, the four numbers of the aligned motion's last
# frame the next step is aligned to (MG_CONSTRAINT_VALUE_*): the chain never leaves the device arithmetic, the host only
# slices.  Alignment itself is PARITY UNPINNED (anim_utils absent), as for the per-primitive objectives.
# ---------------------------------------------------------------------------------------------------------------------
def _aligning_node(motion_primitive_graph):
    sk = getattr(motion_primitive_graph, "skeleton", None)
    node = getattr(sk, "aligning_root_node", None)
    ref_dir = tuple(float(v) for v in getattr(sk, "aligning_root_dir", (0.0, 0.0, 1.0)))
    if node is not None and node == getattr(sk, "root", None):
        node = None
    return node, ref_dir


def _global_blocks(s, motion_primitive_graph, graph_walk_steps, prev_frames, exit_from="frames"):
    """Per step the (n, n_residuals_i) matrix of a batch of concatenated latent vectors, every candidate's steps chained.
    exit_from: "frames" -- the next step is aligned to the last FRAME of this step's aligned motion (obj_global_error_sum,
    obj_global_residual_vector: get_motion_vector()[-1], canonical time F); "coeffs" -- to its last CONTROL POINT
    (obj_global_residual_vector_and_naturalness hands `.coeffs` on, objective_functions.py:373: the spline at its last knot)."""
    S, single = _batch(s)
    S = np.asarray(S, dtype=np.float64)
    n = len(S)
    hip_sk = getattr(motion_primitive_graph, "hip_skeleton", None)
    node_name, ref_dir = _aligning_node(motion_primitive_graph)
    if node_name is not None and hip_sk is None:
        raise NotImplementedError("aligning node %r is not the root joint: the graph needs a _capi.Skeleton as .hip_skeleton" % (node_name,))
    joint = 0 if node_name is None else node_name
    state, offset, blocks = None, 0, []
    for step in graph_walk_steps:
        node = motion_primitive_graph.nodes[step.node_key]
        prim = _prim_of(node)
        Li = int(step.n_spatial_components)
        alpha = np.ascontiguousarray(S[:, offset:offset + Li])
        offset += Li
        cons = step.motion_primitive_constraints
        keyframes, trajectories = split_trajectories(constraints_to_device_form(_constraint_list(cons)))
        if trajectories:
            raise NotImplementedError("trajectory constraints inside a chained graph-walk objective")
        sk = getattr(cons, "hip_skeleton", None) or hip_sk
        F = float(prim.n_canonical_frames)
        t_exit = F if exit_from == "frames" else F - 1.0
        exits = [{"type": "value_heading", "t": t_exit, "weight": 1.0, "axis": 0, "joint": joint, "ref_dir": ref_dir},
                 {"type": "value_heading", "t": t_exit, "weight": 1.0, "axis": 2, "joint": joint, "ref_dir": ref_dir},
                 {"type": "value_position", "t": t_exit, "weight": 1.0, "axis": 0},
                 {"type": "value_position", "t": t_exit, "weight": 1.0, "axis": 2}]
        local = bool(getattr(cons, "is_local", False))
        if state is None:
            # first step: the walk's previous frames (or start pose, or nothing) -- the same record for every candidate
            al_motion = alignment_from_prev_frames(prev_frames, type("_NotLocal", (), {"start_pose": getattr(cons, "start_pose", None), "is_local": False,
                                                                                      "skeleton": getattr(cons, "skeleton", getattr(motion_primitive_graph, "skeleton", None))})(), sk)
            al_cons = None if local else al_motion
            if al_cons is al_motion:
                res = prim.score_constraint_residuals(cached_constraint_set(prim, keyframes + exits, sk, al_motion), alpha)
                res_k, new_state = res[:, :len(keyframes)], res[:, len(keyframes):]
            else:
                res_k = prim.score_constraint_residuals(cached_constraint_set(prim, keyframes, sk, None), alpha) if keyframes else np.zeros((n, 0))
                new_state = prim.score_constraint_residuals(cached_constraint_set(prim, exits, sk, al_motion), alpha)
        else:
            template = {"joint": joint, "position": (0.0, 0.0, 0.0), "heading": (0.0, 1.0), "ref_dir": ref_dir}
            if local:
                res_k = prim.score_constraint_residuals(cached_constraint_set(prim, keyframes, sk, None), alpha) if keyframes else np.zeros((n, 0))
                new_state = prim.score_constraint_residuals_chained(cached_constraint_set(prim, exits, sk, template), alpha, state)
            else:
                res = prim.score_constraint_residuals_chained(cached_constraint_set(prim, keyframes + exits, sk, template), alpha, state)
                res_k, new_state = res[:, :len(keyframes)], res[:, len(keyframes):]
        blocks.append(group_residuals(keyframes, res_k) if keyframes else res_k)
        state = np.ascontiguousarray(new_state)
        if hasattr(cons, "evaluations"):
            cons.evaluations += n
    return S, single, blocks


def obj_global_error_sum(s, data):
    """objective_functions.py:290-316: the sum over the walk's steps of MotionPrimitiveConstraints.evaluate, each step scored
    against the previous step's aligned frames -> (n,) (float for 1-D s).  (The reference prints the value on every call.)"""
    motion_primitive_graph, graph_walk_steps, error_scale, quality_scale, prev_frames = data
    S, single, blocks = _global_blocks(s, motion_primitive_graph, graph_walk_steps, prev_frames, "frames")
    err = np.zeros(len(S))
    for b in blocks:
        err = err + b.sum(axis=1)
    return float(err[0]) if single else err


def obj_global_residual_vector(s, data):
    """objective_functions.py:319-345: the steps' residual vectors (each zero-padded to its number of variables) side by side,
    divided by init_error_sum.  NB the reference hands obj_spatial_error_residual_vector a FIVE-element step_data where that
    function unpacks six (:341 vs :220), so the reference raises ValueError here; this is the evident intent, with the missing
    per-step init_error_sum = 1 as in the _and_naturalness form (:369)."""
    motion_primitive_graph, graph_walk_steps, error_scale, quality_scale, prev_frames, init_error_sum = data
    S, single, blocks = _global_blocks(s, motion_primitive_graph, graph_walk_steps, prev_frames, "frames")
    cols = [_pad(b, int(step.n_spatial_components)) for b, step in zip(blocks, graph_walk_steps)]
    out = np.hstack(cols) / init_error_sum
    return out[0] if single else out


def obj_global_residual_vector_and_naturalness(s, data):
    """objective_functions.py:348-380: per step residual_i * error_scale - log p(concat(alpha, the step's time latents)) *
    quality_scale, zero-padded to the step's FULL number of latents, side by side, divided by init_error_sum; the next step is
    aligned to this step's aligned CONTROL POINTS (the reference passes `.coeffs` on as frames, :362,:373)."""
    motion_primitive_graph, graph_walk_steps, error_scale, quality_scale, prev_frames, init_error_sum = data
    S, single, blocks = _global_blocks(s, motion_primitive_graph, graph_walk_steps, prev_frames, "coeffs")
    cols, offset = [], 0
    for b, step in zip(blocks, graph_walk_steps):
        Li = int(step.n_spatial_components)
        node = motion_primitive_graph.nodes[step.node_key]
        prim = _prim_of(node)
        tail = np.asarray(step.parameters, dtype=np.float64)[Li:]
        concat = np.hstack([S[:, offset:offset + Li], np.tile(tail, (len(S), 1))]) if len(tail) else S[:, offset:offset + Li]
        offset += Li
        nll = -prim.gmm_log_prob(np.ascontiguousarray(concat, dtype=np.float64)) * quality_scale
        cols.append(_pad(b * error_scale + nll[:, None], concat.shape[1]))
    out = np.hstack(cols) / init_error_sum
    return out[0] if single else out


# ---------------------------------------------------------------------------------------------------------------------
# Time constraints (reference constraints/time_constraints.py:25-110, objective optimization/objective_functions.py:270-287)
# ---------------------------------------------------------------------------------------------------------------------
class HipTimeConstraints(object):
    """TimeConstraints for batches: `s` concatenates the TIME latents of the steps start_step .. end_step; per step the
    canonical time functions of all candidates come from one launch (mg_time_function_canonical), their inversion runs on the
    host with scipy as in the reference, the naturalness term is one mixture launch per step."""

    def __init__(self, motion_primitive_graph, graph_walk, start_step, end_step, constraint_list):
        self.start_step, self.end_step, self.constraint_list = start_step, end_step, constraint_list
        self.start_keyframe = self._get_start_frame(motion_primitive_graph, graph_walk, start_step)

    def _get_start_frame(self, graph, graph_walk, start_step):
        start_keyframe = 0
        for i in range(0, max(start_step, 0)):
            tf = graph.nodes[graph_walk.steps[i].node_key].back_project_time_function(graph_walk.steps[i].parameters)
            start_keyframe += tf[-1]
        return start_keyframe

    def _time_functions(self, S, graph, graph_walk):
        """[step][candidate] -> time function t'(t) (1-D array)"""
        out, offset = [], 0
        for step in graph_walk.steps[self.start_step:self.end_step]:
            nt = int(step.n_time_components)
            base = np.asarray(step.parameters, dtype=np.float64)
            vecs = np.tile(base, (len(S), 1))
            vecs[:, int(step.n_spatial_components):] = S[:, offset:offset + nt]
            offset += nt
            node = graph.nodes[step.node_key]
            # the wrapper's back_project_time_function (motion_primitive_wrapper.py:233-249, legacy branch) for every candidate:
            # one launch where the node offers the batched form
            out.append(node.back_project_time_functions(vecs) if hasattr(node, "back_project_time_functions")
                       else [node.back_project_time_function(v) for v in vecs])
        return out

    def evaluate_graph_walk(self, s, graph, graph_walk):
        """time_constraints.py:40-50,68-91 for every candidate -> (n,)"""
        S, single = _batch(s)
        S = np.asarray(S, dtype=np.float64)
        tfs = self._time_functions(S, graph, graph_walk)
        frame_time = graph.skeleton.frame_time
        err = np.zeros(len(S))
        for b in range(len(S)):
            for step_index, keyframe_index, desired_time in self.constraint_list:
                n_frames, e = self.start_keyframe, 10000.0        # (the reference's value when the constrained step is beyond the walk)
                for k, per_step in enumerate(tfs):
                    tf = per_step[b]
                    if k < step_index:
                        n_frames += tf[-1]
                    else:
                        if keyframe_index >= len(tf):
                            e = 0.0
                        else:
                            n_frames += int(tf[keyframe_index]) + 1
                            e = (desired_time - n_frames * frame_time) ** 2
                        break
                err[b] += e
        return float(err[0]) if single else err

    def get_average_loglikelihood(self, s, graph, graph_walk):
        """time_constraints.py:93-102 for every candidate -> (n,)"""
        S, single = _batch(s)
        S = np.asarray(S, dtype=np.float64)
        total, count, offset = np.zeros(len(S)), 0, 0
        for step in graph_walk.steps[self.start_step:self.end_step]:
            nt, ns = int(step.n_time_components), int(step.n_spatial_components)
            X = np.hstack([np.tile(np.asarray(step.parameters, dtype=np.float64)[:ns], (len(S), 1)), S[:, offset:offset + nt]])
            offset += nt
            total += _prim_of(graph.nodes[step.node_key]).gmm_log_prob(np.ascontiguousarray(X))
            count += 1
        out = total / max(count, 1)
        return float(out[0]) if single else out

    def get_initial_guess(self, graph_walk):
        parameters = []
        for step in graph_walk.steps[self.start_step:self.end_step]:
            parameters += np.asarray(step.parameters)[int(step.n_spatial_components):].tolist()
        return parameters


def obj_time_error_sum(s, data):
    """objective_functions.py:270-287: error_scale * time_error + quality_scale * (-average log-likelihood) -> (n,)"""
    motion_primitive_graph, graph_walk, time_constraints, error_scale, quality_scale = data
    time_error = time_constraints.evaluate_graph_walk(s, motion_primitive_graph, graph_walk)
    nll = -np.asarray(time_constraints.get_average_loglikelihood(s, motion_primitive_graph, graph_walk))
    out = error_scale * np.asarray(time_error) + nll * quality_scale
    return float(out) if np.ndim(out) == 0 else out


# ---------------------------------------------------------------------------------------------------------------------
# Step goals (reference optimization/objective_functions.py:59-140).  The reference reads the LAST control point's root position
# straight from the projection (`sfpca.project(s)[idx:idx+3]`, idx = (n_coeffs - 1) * n_dims) of an MGRD model; here the same
# three numbers are rows of E' and mean' of the legacy primitive: pos = E_last . s + mean_last, y dropped.
# ---------------------------------------------------------------------------------------------------------------------
def _last_control_point_rows(motion_primitive):
    mp = motion_primitive.motion_primitive if hasattr(motion_primitive, "motion_primitive") else motion_primitive
    sp = mp.s_pca
    NB, D = int(sp["n_basis"]), int(sp["n_dim"])
    E = np.asarray(sp["eigen_vectors"], dtype=np.float64)            # (NB * D, L) as loaded (motion_primitive.py:156)
    mean = np.asarray(sp["mean_vector"], dtype=np.float64)
    scale = np.asarray(getattr(mp, "translation_maxima", (1.0, 1.0, 1.0)), dtype=np.float64)
    idx = (NB - 1) * D
    return E[idx:idx + 3] * scale[:, None], mean[idx:idx + 3] * scale


def _step_goal(s, data):
    motion_primitive, mp_constraints = data[0], data[1]
    target = np.asarray(_constraint_list(mp_constraints)[0].position, dtype=np.float64)
    S, single = _batch(s)
    S = np.asarray(S, dtype=np.float64)
    E3, m3 = _last_control_point_rows(motion_primitive)
    n_spatial = E3.shape[1]
    pos = S[:, :n_spatial] @ E3.T + m3
    pos[:, 1] = 0.0
    delta = target[None, :] - pos
    return S, single, E3, delta, n_spatial


def step_goal_error(s, data):
    """objective_functions.py:59-71 -> (n,)"""
    S, single, E3, delta, _ = _step_goal(s, data)
    err = np.einsum("ij,ij->i", delta, delta)
    return float(err[0]) if single else err


def step_goal_jac(s, data):
    """objective_functions.py:73-90 -> (n, len(s)): 2 * E_last^T (pos - target) on the spatial latents, zeros elsewhere"""
    S, single, E3, delta, n_spatial = _step_goal(s, data)
    jac = np.zeros_like(S)
    jac[:, :n_spatial] = 2.0 * (-delta) @ E3
    return jac[0] if single else jac


def step_goal_and_naturalness(s, data):
    """objective_functions.py:94-107: step_goal_error - log p(s)"""
    S, single, E3, delta, _ = _step_goal(s, data)
    err = np.einsum("ij,ij->i", delta, delta) - _prim_of(data[0]).gmm_log_prob(np.ascontiguousarray(S))
    return float(err[0]) if single else err


def step_goal_and_naturalness_jac(s, data):
    """objective_functions.py:122-140: step_goal_jac - log_likelihood_jac"""
    S, single = _batch(s)
    jac = np.atleast_2d(step_goal_jac(S, data)) - np.atleast_2d(log_likelihood_jac(S, data[0]))
    return jac[0] if single else jac
