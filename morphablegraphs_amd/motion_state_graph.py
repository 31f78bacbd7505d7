"""A minimal container for BASELINE.json config 3: the ~16 primitives of a MotionStateGraph resident on
one GPU, and the batched form of GraphWalkPlanner's option evaluation
(reference morphablegraphs/motion_generator/graph_walk_planner.py:184-226): for every outgoing option draw
n candidates, score them against the path-following constraints, keep the first minimum per option, pick
the option with the smallest error (np.argmin over options, graph_walk_planner.py:191-192).
Graph loading / transitions / control flow stay in the reference; only the scoring is replaced."""
import numpy as np

from . import _capi
from .candidate_scoring import constraints_to_device_form
from .motion_primitive import HipMotionPrimitive, get_context


class HipPrimitiveSet(object):
    def __init__(self, primitives_json, context=None, device=0):
        self.ctx = context or get_context(device)
        self.nodes = {}
        for data in primitives_json:
            p = HipMotionPrimitive(None, context=self.ctx)
            p._initialize_from_json(data)
            self.nodes[p.name] = p

    def evaluate_options(self, options, constraints_per_option, n_samples, rng_seed=None):
        """options: node names; constraints_per_option: name -> constraint list.  Returns
        (best_option, {name: (best_sample, min_error)})."""
        results = {}
        for name in options:
            node = self.nodes[name]
            prim = node._prim
            if rng_seed is not None:
                np.random.seed(rng_seed)
            samples = node.sample_low_dimensional_vector(n_samples)
            cset = _capi.ConstraintSet(prim, constraints_to_device_form(constraints_per_option[name]))
            try:
                S = _capi._latents(samples)
                d_s = self.ctx.upload(S)
                d_e = self.ctx.malloc(len(S) * 8)
                prim.score_constraints_dev(cset, d_s, S.dtype, len(S), S.shape[1], d_e, np.float64)
                idx, err = self.ctx.argmin_first(d_e, len(S), np.float64)
                d_s.free()
                d_e.free()
            finally:
                cset.close()
            results[name] = (samples[idx], err)
        errors = [results[n][1] for n in options]
        return options[int(np.argmin(errors))], results
