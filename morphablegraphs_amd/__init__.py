"""MI355X-native motion-primitive back-projection / GMM scoring behind the morphablegraphs plugin surface.
Importing the package does not load the HIP library; the first backend object does (and fails loudly
if libmg_hip.so or a gfx950 device is missing -- there is no CPU fallback)."""
from .gaussian_mixture import HipGaussianMixture, sample_like_sklearn  # noqa: F401
from .motion_primitive import HipMotionPrimitive, get_context  # noqa: F401
from .motion_primitive_wrapper import (HipMotionPrimitiveModelWrapper, HipStaticMotionPrimitive,  # noqa: F401
                                       mgrd_json_to_legacy)
from .motion_spline import HipMotionSpline  # noqa: F401

__all__ = ["HipGaussianMixture", "HipMotionPrimitive", "HipMotionPrimitiveModelWrapper", "HipMotionSpline",
           "HipStaticMotionPrimitive", "get_context", "mgrd_json_to_legacy", "sample_like_sklearn"]
