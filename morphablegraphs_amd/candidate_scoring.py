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


_CSET_CACHE = []   # [(key, ConstraintSet)], most recent last: an optimizer calls the objective hundreds of times with
_CSET_CACHE_SIZE = 64   # the same constraints, and building a set uploads its fused keyframe matrices


def _constraint_key(prim, clist, skeleton):
    def freeze(v):
        if isinstance(v, dict):
            return tuple(sorted((k, freeze(x)) for k, x in v.items()))
        if isinstance(v, (list, tuple, np.ndarray)):
            return tuple(freeze(x) for x in v)
        return v
    return (id(prim), prim.handle.value, id(skeleton), freeze(clist))


def cached_constraint_set(prim, clist, skeleton=None):
    """A device constraint set for these (device-form) constraints, reused across calls: a graph walk or an
    optimizer evaluates the same constraints again and again, and building a set uploads its fused matrices."""
    key = _constraint_key(prim, clist, skeleton)
    for i in range(len(_CSET_CACHE) - 1, -1, -1):   # entries whose primitive has been closed meanwhile are dropped
        if not (_CSET_CACHE[i][1].handle and _CSET_CACHE[i][1].prim.handle and _CSET_CACHE[i][1].prim.ctx.handle):
            _CSET_CACHE.pop(i)
    for i, (k, cs) in enumerate(_CSET_CACHE):
        if k == key:
            _CSET_CACHE.append(_CSET_CACHE.pop(i))
            return cs
    cs = _capi.ConstraintSet(prim, clist, skeleton)
    _CSET_CACHE.append((key, cs))
    while len(_CSET_CACHE) > _CSET_CACHE_SIZE:
        _CSET_CACHE.pop(0)[1].close()
    return cs


def clear_constraint_cache():
    """Drop the cached device constraint sets (call before closing a primitive they belong to)."""
    while _CSET_CACHE:
        _CSET_CACHE.pop()[1].close()



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
    cset = cached_constraint_set(prim, constraints_to_device_form(clist), skeleton)
    best_idx, min_error = prim.best_candidate(cset, samples)   # one upload, two launches, 16 bytes back
    if hasattr(constraints, "min_error"):
        constraints.min_error = min_error
    if hasattr(constraints, "evaluations"):
        constraints.evaluations += len(samples)
    return samples[best_idx], min_error


def sample_and_evaluate_on_device(mp_node, constraints, n_samples, seed, skeleton=None, dtype=np.float32):
    """The gpu_batch step without the host round trip: component counts from NumPy's global stream (the first
    draw sklearn's GaussianMixture.sample makes), latents from the device Philox sampler (NOT sklearn's Mersenne
    stream: distributional parity only), scoring and first-minimum argmin on the device; only the winning latent
    vector comes back.  Returns (best_sample, min_error)."""
    prim_obj = mp_node.motion_primitive if hasattr(mp_node, "motion_primitive") else mp_node
    prim = prim_obj._prim
    ctx = prim.ctx
    clist = constraints.constraints if hasattr(constraints, "constraints") else constraints
    skeleton = skeleton if skeleton is not None else getattr(constraints, "hip_skeleton", None)
    cset = cached_constraint_set(prim, constraints_to_device_form(clist), skeleton)
    L = prim.n_components
    weights = np.asarray(prim_obj.gaussian_mixture_model.weights_, dtype=np.float64)
    counts = np.random.multinomial(int(n_samples), weights / weights.sum()).astype(np.int64)
    item = np.dtype(dtype).itemsize
    d_x = ctx.malloc(max(int(n_samples), 1) * L * item)
    try:
        prim.gmm_sample_dev(counts, seed, d_x, dtype, L)
        best_idx, min_error = prim.best_candidate_dev(cset, d_x, dtype, int(n_samples), L)
        best = ctx.download(d_x.ptr.value + best_idx * L * item, (L,), dtype)
    finally:
        d_x.free()
    if hasattr(constraints, "min_error"):
        constraints.min_error = min_error
    if hasattr(constraints, "evaluations"):
        constraints.evaluations += int(n_samples)
    return best.astype(np.float64), min_error
