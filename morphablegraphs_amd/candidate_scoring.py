"""Batched candidate scoring: the GPU replacement of the per-sample loop in
MotionPrimitiveGenerator.evaluate_samples_using_constraints
(reference morphablegraphs/motion_generator/motion_primitive_generator.py:230-261), shaped after the
MGRD precedent MGRDSampleFilter.score_samples (reference mgrd_sample_filter.py:63-75).

Constraints covered by the fused kernel: the FK-free ones path following uses
(locomotion_constraints_builder.py:82-117) -- root position at a canonical keyframe
(GlobalTransformConstraint, position only, root joint) and 2-D heading (Direction2DConstraint) -- and, given
a `_capi.Skeleton` (argument, or `hip_skeleton` attribute of the constraints object), the position of any
other joint by forward kinematics (hands, feet).  Anything else raises -- there is no silent CPU fallback.
"""
import numpy as np

from . import _capi

SAMPLING_MODE_GPU_BATCH = "gpu_batch"   # a 4th constrained_sampling_mode next to the reference's three


def constraints_to_device_form(constraints, root_joint=None):
    """Accepts either ready dicts {"type","t","weight","target"[,"ref_dir"]} or reference-shaped
    constraint objects (attributes canonical_keyframe, weight_factor, and position / target_dir)."""
    out = []
    for c in constraints:
        if isinstance(c, dict):
            out.append(c)
            continue
        t = float(c.canonical_keyframe)
        w = float(getattr(c, "weight_factor", 1.0))
        if hasattr(c, "target_dir"):
            rd = getattr(getattr(c, "skeleton", None), "aligning_root_dir", (0.0, 0.0, 1.0))
            out.append({"type": "direction", "t": t, "weight": w, "target": [float(c.target_dir[0]), float(c.target_dir[1])],
                        "ref_dir": tuple(float(v) for v in rd)})
        elif getattr(c, "position", None) is not None and getattr(c, "orientation", None) is None:
            joint = getattr(c, "joint_name", root_joint)
            if root_joint is not None and joint != root_joint:
                # any other joint goes through the forward-kinematics constraint (needs a skeleton on the set)
                out.append({"type": "joint_position", "t": t, "weight": w, "target": list(c.position), "joint": joint})
            else:
                out.append({"type": "position", "t": t, "weight": w, "target": list(c.position)})
        else:
            raise NotImplementedError("constraint %r is not covered by the fused GPU scorer" % (type(c).__name__,))
    return out


class HipSampleFilter(object):
    """score_samples(primitive, samples, constraints) -> errors[n], like MGRDSampleFilter."""

    @staticmethod
    def score_samples(motion_primitive, samples, constraints, dtype=np.float64, skeleton=None):
        prim = motion_primitive._prim if hasattr(motion_primitive, "_prim") else motion_primitive
        cset = _capi.ConstraintSet(prim, constraints_to_device_form(constraints), skeleton)
        try:
            return prim.score_constraints(cset, np.asarray(samples), dtype=dtype)
        finally:
            cset.close()


def evaluate_samples_using_constraints(samples, mp_node, constraints, prev_frames=None, skeleton=None):
    """Drop-in for MotionPrimitiveGenerator.evaluate_samples_using_constraints in local-coordinate
    mode: returns (samples[best_idx], min_error) with the reference's first-minimum rule, and
    updates constraints.min_error / constraints.evaluations when those attributes exist."""
    if prev_frames is not None:
        raise NotImplementedError("global-coordinate scoring needs anim_utils' alignment; use use_local_coordinates")
    samples = np.asarray(samples)
    prim_obj = mp_node.motion_primitive if hasattr(mp_node, "motion_primitive") else mp_node
    prim = prim_obj._prim
    clist = constraints.constraints if hasattr(constraints, "constraints") else constraints
    skeleton = skeleton if skeleton is not None else getattr(constraints, "hip_skeleton", None)
    cset = _capi.ConstraintSet(prim, constraints_to_device_form(clist), skeleton)
    ctx = prim.ctx
    try:
        S = _capi._latents(samples)
        d_s = ctx.upload(S)
        d_e = ctx.malloc(max(len(S), 1) * 8)
        prim.score_constraints_dev(cset, d_s, S.dtype, len(S), S.shape[1], d_e, np.float64)
        best_idx, min_error = ctx.argmin_first(d_e, len(S), np.float64)
        d_s.free()
        d_e.free()
    finally:
        cset.close()
    if hasattr(constraints, "min_error"):
        constraints.min_error = min_error
    if hasattr(constraints, "evaluations"):
        constraints.evaluations += len(samples)
    return samples[best_idx], min_error
