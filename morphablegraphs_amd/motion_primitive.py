"""MotionPrimitive duck type backed by libmg_hip.so.

Mirror of reference morphablegraphs/motion_model/motion_primitive.py:41-380: same attribute names
(``s_pca``, ``t_pca``, ``gaussian_mixture_model``, ``translation_maxima``,
``n_canonical_frames`` ...), same methods and error behaviour, so
``MotionPrimitiveModelWrapper.motion_primitive`` can be swapped without touching
``motion_generator``.  Single-sample calls return float64 results from the float64 kernels;
the ``*_batch`` methods are the hot path (float32 frames, MFMA kernel).
"""
import json

import numpy as np

from . import _capi
from .gaussian_mixture import HipGaussianMixture
from .motion_spline import HipMotionSpline

_default_context = {}


def get_context(device=0):
    """One libmg_hip context per (process, device)."""
    ctx = _default_context.get(device)
    if ctx is None or ctx.handle is None:
        ctx = _capi.Context(device)
        _default_context[device] = ctx
    return ctx


class HipMotionPrimitive(object):
    def __init__(self, filename=None, context=None, device=0):
        self.filename = filename
        self.name = ""
        self.gaussian_mixture_model = None
        self.s_pca = dict()
        self.t_pca = dict()
        self.n_canonical_frames = 0
        self.translation_maxima = np.array([1.0, 1.0, 1.0])
        self.smooth_time_parameters = False
        self.has_time_parameters = True
        self.has_semantic_parameters = False
        self.animated_joints = []
        self._ctx = context
        self._device = device
        self._prim = None
        if self.filename is not None:
            self._load(self.filename)

    # ---- loading (reference motion_primitive.py:79-124) ------------------------------------
    def _load(self, filename=None):
        with open(filename, "r") as infile:
            self._initialize_from_json(json.load(infile))

    def _initialize_from_json(self, data):
        if "name" in data:
            self.name = data["name"]
        if "semantic_label" in data:
            self.has_semantic_parameters = True
            self.semantic_labels = data["semantic_label"]
        self.n_canonical_frames = data["n_canonical_frames"]
        self.canonical_time_range = np.arange(0, self.n_canonical_frames)
        if self._ctx is None:
            self._ctx = get_context(self._device)
        self._prim = _capi.Primitive(self._ctx, data)
        self._init_gmm_from_json(data)
        self._init_spatial_parameters_from_json(data)
        if "eigen_vectors_time" in data:
            self._init_time_parameters_from_json(data)
            self.has_time_parameters = True
        else:
            self.has_time_parameters = False
            self.t_pca = dict()
            self.t_pca["n_components"] = 0
        if "animated_joints" in data:
            self.animated_joints = data["animated_joints"]

    def _init_gmm_from_json(self, data):
        self.gaussian_mixture_model = HipGaussianMixture(self._prim, data["gmm_weights"], data["gmm_means"],
                                                         data["gmm_covars"])

    def _init_spatial_parameters_from_json(self, data):
        self.translation_maxima = np.array(data["translation_maxima"])
        self.s_pca = dict()
        self.s_pca["eigen_vectors"] = np.transpose(np.array(data["eigen_vectors_spatial"]))
        self.s_pca["mean_vector"] = np.array(data["mean_spatial_vector"])
        self.s_pca["n_basis"] = int(data["n_basis_spatial"])
        self.s_pca["n_dim"] = int(data["n_dim_spatial"])
        self.s_pca["n_components"] = len(self.s_pca["eigen_vectors"].T)
        self.s_pca["knots"] = np.asarray(data["b_spline_knots_spatial"])

    def _init_time_parameters_from_json(self, data):
        # reference motion_primitive.py:164-181
        self.t_pca = dict()
        self.t_pca["eigen_vectors"] = np.array(data["eigen_vectors_time"])
        self.t_pca["mean_vector"] = np.array(data["mean_time_vector"])
        self.t_pca["n_basis"] = int(data["n_basis_time"])
        self.t_pca["n_dim"] = 1
        self.t_pca["n_components"] = len(self.t_pca["eigen_vectors"].T)
        self.t_pca["knots"] = np.asarray(data["b_spline_knots_time"])
        self.t_pca["eigen_coefs"] = list(zip(*self.t_pca["eigen_vectors"]))

    # ---- sampling (reference motion_primitive.py:182-204) -----------------------------------
    def sample_low_dimensional_vector(self, n_samples=1):
        assert self.gaussian_mixture_model is not None, "Motion primitive not initialized."
        return self.gaussian_mixture_model.sample(n_samples)[0]

    def sample(self, use_time_parameters=True):
        return self.back_project(np.ravel(self.sample_low_dimensional_vector()), use_time_parameters)

    # ---- back projection (reference motion_primitive.py:206-256) ---------------------------
    def _strip_semantic_label(self, s):
        semantic_annotation = None
        if self.has_semantic_parameters:
            semantic_label = s[-1]
            for key, value in self.semantic_labels.items():
                if int(np.round(semantic_label)) == value:
                    semantic_annotation = key
            if semantic_annotation is None:
                raise ValueError('Unknown semantic label!')
            s = np.delete(s, -1)
        return s, semantic_annotation

    def back_project(self, s, use_time_parameters=True, speed=1.0):
        s = np.asarray(s, dtype=np.float64)
        s, semantic_annotation = self._strip_semantic_label(s)
        spatial_coeffs = self.back_project_spatial_coeffs(s[:self.s_pca["n_components"]])
        if self.has_time_parameters and use_time_parameters:
            time_function = self.back_project_time_function(s[self.s_pca["n_components"]:], speed)
        else:
            time_function = np.linspace(0, self.n_canonical_frames, int(self.n_canonical_frames * (1.0 / speed)))
        return HipMotionSpline(spatial_coeffs, time_function, self.s_pca["knots"], semantic_annotation,
                               low_dimensional_parameters=s, primitive=self._prim)

    def back_project_spatial_coeffs(self, alpha):
        alpha = np.asarray(alpha, dtype=np.float64).reshape(1, -1)
        return self._prim.back_project_coeffs(alpha, dtype=np.float64)[0]

    # ---- time warp (reference motion_primitive.py:258-331) -----------------------------------------------
    def _mean_temporal(self):
        """The mean time spline at the canonical frames (motion_primitive.py:258-266); host, scipy, as the reference."""
        import scipy.interpolate as si
        return si.splev(self.canonical_time_range, (self.t_pca["knots"], self.t_pca["mean_vector"], 3))

    def _back_transform_gamma_to_canonical_time_function(self, gamma):
        """t(t') at the canonical frames: cumulative sum of exp(mean + harmonics . gamma), minus 1
        (motion_primitive.py:289-302), on the device; see also the *_batch variant."""
        return self._prim.time_function_canonical(np.asarray(gamma, dtype=np.float64).reshape(1, -1))[0]

    def back_transform_gamma_to_canonical_time_function_batch(self, gammas):
        """(B, n_time_components) -> (B, n_canonical_frames) float64 in one launch."""
        return self._prim.time_function_canonical(gammas)

    def _invert_canonical_to_sample_time_function(self, canonical_time_function, speed=1.0):
        """t'(t) from t(t') by an interpolating cubic spline through (t(t'), t') sampled at the integer sample times
        (motion_primitive.py:304-319), with scipy's splrep / splev on the host exactly as the reference does.
        PARITY UNPINNED for the sample count: the reference passes the FLOAT `num` to np.linspace, which raises
        TypeError on every NumPy >= 1.18; this restatement truncates it with int() -- the value
        NumPy < 1.18 silently used."""
        from scipy.interpolate import splev, splrep
        F = self.n_canonical_frames
        t_of_tprime = np.asarray(canonical_time_function, dtype=np.float64)
        # the inverse map as an interpolating cubic: knots at the canonical function's values, ordinates the sample indices
        inverse = splrep(t_of_tprime, np.arange(F), k=3)
        last = t_of_tprime[-2]
        n_inner = int(np.round(last) * (1.0 / speed))      # the product, as the reference forms it (x * (1 / speed) and x / speed can differ in the last bit)
        inner = splev(np.linspace(1.0, last, n_inner), inverse)
        # the reference pins both ends: sample time 0 maps to canonical 0, the last one to the last canonical frame
        return np.concatenate(([0.0], inner, [F - 1.0]))

    def _smooth_time_function(self, time_function):
        from scipy.signal import savgol_filter        # motion_primitive.py:321-331
        return np.array(savgol_filter(time_function, 15, 3))

    def back_project_time_function(self, gamma, speed=1.0):
        """The time-warp t'(t) of a sample (motion_primitive.py:268-287), canonical time function and its inversion on the
        device (mg_time_function_sample; the host route with scipy, _invert_canonical_to_sample_time_function, stays for
        callers that hold a canonical time function of their own)."""
        times, lens = self._prim.time_function_sample(np.asarray(gamma, dtype=np.float64).reshape(1, -1), speed)
        sample_time_function = times[0, :lens[0]].copy()
        if self.smooth_time_parameters:
            return self._smooth_time_function(sample_time_function)
        return sample_time_function

    def back_project_warped_batch(self, samples, speed=1.0, dtype=np.float64):
        """back_project(s, use_time_parameters=True).get_motion_vector() for every row of `samples` (n, n_spatial + n_time
        latents) in two launches -- what GraphWalk.convert_graph_walk_to_quaternion_frames does step by step
        (graph_walk.py:154-176).  Returns (frames (n, t_max, D) padded with NaN, lengths (n), times (n, t_max) padded with NaN).
        With smooth_time_parameters the time functions pass through the host (savgol_filter, as in the reference)."""
        S = np.ascontiguousarray(np.asarray(samples, dtype=np.float64))
        n_s = self.s_pca["n_components"]
        if not self.has_time_parameters:
            raise ValueError("the primitive has no time parameters")
        times, lens = self._prim.time_function_sample(S[:, n_s:], speed)
        t_max = int(lens.max()) if len(lens) else 0
        times = np.ascontiguousarray(times[:, :t_max])
        if self.smooth_time_parameters:
            for b in range(len(S)):
                times[b, :lens[b]] = self._smooth_time_function(times[b, :lens[b]])
        frames = self._prim.back_project_frames_at(S[:, :n_s], times, lens, dtype=dtype)
        return frames, lens, times

    # ---- batched hot path ------------------------------------------------------------------------
    def back_project_frames_batch(self, samples, times=None):
        """(B, L) latents -> (B, F, D) float32 frames: back_project(s, False).get_motion_vector() for every row
        in one launch (or .evaluate(times) when times is given)."""
        S = np.asarray(samples)
        S = S[:, :self.s_pca["n_components"]]
        if times is None:
            return self._prim.back_project_frames(S)
        g = self._prim.time_grid(times)
        try:
            return self._prim.back_project_frames(S, g)
        finally:
            g.close()

    def back_project_spatial_coeffs_batch(self, samples, dtype=np.float64):
        return self._prim.back_project_coeffs(np.asarray(samples)[:, :self.s_pca["n_components"]], dtype=dtype)

    def score_samples_batch(self, samples, dtype=np.float64):
        """log p(x) of every row under the primitive's mixture."""
        return self._prim.gmm_log_prob(np.asarray(samples), dtype=dtype)

    # ---- getters (reference motion_primitive.py:333-380) ---------------------------------------
    def get_n_canonical_frames(self):
        return self.n_canonical_frames

    def get_n_spatial_components(self):
        return self.s_pca["n_components"]

    def get_n_time_components(self):
        if "n_components" in self.t_pca.keys():
            return self.t_pca["n_components"]
        return 0

    def get_spatial_jacobian(self):
        return self.s_pca["eigen_vectors"].T

    def get_animated_joints(self):
        return self.animated_joints

    def path_following_obj(self, target, alpha):
        coeffs = self.back_project_spatial_coeffs(alpha)
        return np.linalg.norm(target - coeffs[-1, :3])
