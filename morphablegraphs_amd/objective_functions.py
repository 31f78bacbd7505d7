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


def obj_spatial_error_sum_and_naturalness(s, data):
    """objective_functions.py:162-184: error_scale * spatial_error + quality_scale * (-log p(s)).
    (The reference function computes this value and then falls off its end without `return`, so scipy receives
    None there; the batched form returns the value it computes.)"""
    motion_primitive, mp_constraints, prev_frames, error_scale, quality_scale = data[0], data[1], data[2], data[-3], data[-2]
    S, single = _batch(s)
    prim = _prim_of(motion_primitive)
    spatial = _residuals(prim, mp_constraints, S, prev_frames, sums=True)
    _note(mp_constraints, float(spatial[-1]) if len(spatial) else 0.0, len(S))
    err = error_scale * spatial + (-prim.gmm_log_prob(S.astype(np.float64))) * quality_scale
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
