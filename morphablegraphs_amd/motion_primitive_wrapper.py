"""The plugin seam: a MotionPrimitiveModelWrapper whose ``motion_primitive`` is the HIP backend.

Mirror of reference morphablegraphs/motion_model/motion_primitive_wrapper.py:43-375 (legacy,
``self.mgrd == False`` branch): same method names, format dispatch and getters, so
``MotionStateGraphNode`` (which inherits the wrapper, motion_state_graph_node.py:45) and every
``motion_generator`` call site keep working unchanged.  Static primitives
(``"spatial_coeffs"`` files, static_motion_primitive.py:28-76) are constant and stay on the host.
"""
import json

import numpy as np

from .gaussian_mixture import HipGaussianMixture
from .motion_primitive import HipMotionPrimitive
from .motion_spline import HipMotionSpline


def mgrd_json_to_legacy(data):
    """v3 ``sspm/tspm/gmm`` file -> legacy dict, as _load_legacy_model_from_mgrd_json does
    (reference motion_primitive_wrapper.py:87-115): translation_maxima = [1,1,1], no time
    parameters, n_canonical_frames = int(max(tspm.knots) + 1)."""
    sspm, tspm, gmm = data["sspm"], data["tspm"], data["gmm"]
    return {"name": data.get("name", ""),
            "eigen_vectors_spatial": sspm["eigen"], "mean_spatial_vector": sspm["mean"],
            "n_basis_spatial": sspm["n_coeffs"], "n_dim_spatial": sspm["n_dims"],
            "b_spline_knots_spatial": sspm["knots"], "animated_joints": sspm.get("animated_joints", []),
            "gmm_covars": gmm["covars"], "gmm_means": gmm["means"], "gmm_weights": gmm["weights"],
            "n_canonical_frames": int(max(tspm["knots"]) + 1), "translation_maxima": np.array([1, 1, 1])}


class HipStaticMotionPrimitive(object):
    """Constant motion with the primitive interface (reference static_motion_primitive.py:28-76).
    Its spline still evaluates on the GPU through HipMotionSpline when a backend handle exists."""

    def __init__(self):
        self.motion_spline = None
        self.name = ""
        self.has_time_parameters = False
        self.has_semantic_parameters = False
        self.n_canonical_frames = 0
        self.animated_joints = []

    def _initialize_from_json(self, data):
        self.name = data["name"]
        self.spatial_coefs = np.array(data["spatial_coeffs"])
        self.knots = np.array(data["knots"])
        self.n_canonical_frames = data["n_canonical_frames"]
        self.time_function = np.array(list(range(self.n_canonical_frames)))
        self.motion_spline = HipMotionSpline(self.spatial_coefs, self.time_function, self.knots, None)
        self.gmm = None
        if "skeleton" in data:
            self.animated_joints = data["skeleton"]["animated_joints"]

    def sample_low_dimensional_vector(self, use_time_parameters=True):
        return [0]

    def sample(self, use_time_parameters=True):
        return self.motion_spline

    def back_project(self, s, use_time_parameters=True, speed=1.0):
        return self.motion_spline

    def back_project_time_function(self, gamma, speed=1.0):
        return self.time_function

    def get_n_spatial_components(self):
        return 1

    def get_n_time_components(self):
        return 0

    def get_gaussian_mixture_model(self):
        return self.gmm

    def get_n_canonical_frames(self):
        return self.n_canonical_frames

    def get_animated_joints(self):
        return self.animated_joints


class HipMotionPrimitiveModelWrapper(object):
    SPLINE_DEGREE = 3

    def __init__(self, context=None, device=0):
        self.motion_primitive = None
        self.use_mgrd_mixture_model = False
        self.keyframes = dict()
        self.mgrd = False
        self._ctx = context
        self._device = device

    def _load_from_file(self, mgrd_skeleton, file_name, animated_joints=None, use_mgrd_mixture_model=False, scale=None):
        with open(file_name, "r") as f:
            data = json.load(f)
        if data is not None:
            self._initialize_from_json(mgrd_skeleton, data, animated_joints, use_mgrd_mixture_model, scale)

    def _initialize_from_json(self, mgrd_skeleton, data, animated_joints=None, use_mgrd_mixture_model=False, scale=None):
        self.mgrd = False
        if "keyframes" in data:
            self.keyframes = data["keyframes"]
        if "spatial_coeffs" in data:
            self.motion_primitive = HipStaticMotionPrimitive()
            self.motion_primitive._initialize_from_json(data)
            return
        legacy = mgrd_json_to_legacy(data) if "tspm" in data else data
        prim = HipMotionPrimitive(None, context=self._ctx, device=self._device)
        prim._initialize_from_json(legacy)
        self.motion_primitive = prim

    # ---- sampling -----------------------------------------------------------------------------
    def sample(self, use_time=True):
        return self.motion_primitive.sample(use_time)

    def sample_low_dimensional_vector(self):
        return self.motion_primitive.sample_low_dimensional_vector(1)

    def sample_low_dimensional_vectors(self, n_samples=1):
        return self.motion_primitive.sample_low_dimensional_vector(n_samples)

    # ---- back projection ------------------------------------------------------------------------
    def back_project(self, s_vec, use_time_parameters=True, speed=1.0):
        return self.motion_primitive.back_project(s_vec, use_time_parameters, speed)

    def back_project_time_function(self, s_vec):
        """reference motion_primitive_wrapper.py:233-249 (legacy branch): the canonical time function of the sample's time
        latents, or the identity for a primitive without a time model."""
        if self.motion_primitive.has_time_parameters:
            return self.motion_primitive._back_transform_gamma_to_canonical_time_function(
                np.asarray(s_vec, dtype=np.float64)[self.get_n_spatial_components():])
        return list(range(0, self.motion_primitive.n_canonical_frames))

    def back_project_time_functions(self, samples):
        """The same for a batch of samples (n, n_spatial + n_time) in one launch: (n, n_canonical_frames)."""
        samples = np.asarray(samples, dtype=np.float64)
        if self.motion_primitive.has_time_parameters:
            return self.motion_primitive.back_transform_gamma_to_canonical_time_function_batch(samples[:, self.get_n_spatial_components():])
        return np.tile(np.arange(self.motion_primitive.n_canonical_frames, dtype=np.float64), (len(samples), 1))

    # ---- getters ---------------------------------------------------------------------------------
    def get_n_canonical_frames(self):
        return self.motion_primitive.n_canonical_frames

    def get_n_spatial_components(self):
        return self.motion_primitive.get_n_spatial_components()

    def get_n_time_components(self):
        return self.motion_primitive.get_n_time_components()

    def get_gaussian_mixture_model(self):
        if isinstance(self.motion_primitive, HipStaticMotionPrimitive):
            return self.motion_primitive.get_gaussian_mixture_model()
        return self.motion_primitive.gaussian_mixture_model

    def get_spatial_eigen_vectors(self, joints=None, frame_idx=-1):
        return self.motion_primitive.s_pca["eigen_vectors"].T

    def get_time_eigen_vector_matrix(self):
        return self.motion_primitive.t_pca["eigen_vectors"]

    def get_animated_joints(self):
        return self.motion_primitive.get_animated_joints()

    @staticmethod
    def load_mixture_model(data, primitive, use_mgrd=False):
        """sklearn-shaped mixture from JSON (reference motion_primitive_wrapper.py:351-370)."""
        return HipGaussianMixture(primitive, data["gmm_weights"], data["gmm_means"], data["gmm_covars"])
