"""MotionSpline duck type backed by the HIP spline-evaluation kernel.

Mirror of reference morphablegraphs/motion_model/motion_spline.py:33-108: same constructor
arguments, attributes (``coeffs`` is read/WRITE -- callers overwrite it with aligned
coefficients, motion_primitive_constraints.py:113), and methods.  Evaluation runs
``mg_spline_evaluate`` (float64, FITPACK splev semantics incl. ext=0 extrapolation) on the
GPU from the spline's *current* coefficients.
"""
import numpy as np

B_SPLINE_DEGREE = 3


class HipMotionSpline(object):
    def __init__(self, canonical_motion_coeffs, time_function, knots, semantic_annotation=None,
                 low_dimensional_parameters=None, primitive=None):
        self.low_dimensional_parameters = low_dimensional_parameters
        self.time_function = time_function
        self.buffered_frames = None
        self.coeffs = canonical_motion_coeffs
        self.knots = knots
        self.semantic_annotation = semantic_annotation
        self.n_pose_parameters = len(canonical_motion_coeffs[0])
        self.n_max_frame = knots[-1]
        self._prim = primitive          # _capi.Primitive: owns the knot vector on the device

    def _evaluate(self, times, grid=None):
        if self._prim is None:
            raise RuntimeError("HipMotionSpline needs its primitive handle (no CPU fallback)")
        coeffs = np.asarray(self.coeffs, dtype=np.float64)
        if grid is not None:
            return self._prim.spline_evaluate(coeffs, grid)[0]
        g = self._prim.time_grid(times)
        try:
            return self._prim.spline_evaluate(coeffs, g)[0]
        finally:
            g.close()

    def get_motion_vector(self, step_size=None):
        """(n_frames, n_channels) float64 frames on the spline's time function
        (reference motion_spline.py:71-86)."""
        if step_size is not None:
            # the reference passes a float `num` to np.linspace here (motion_spline.py:80-81), a TypeError on
            # every supported numpy; the intended grid is restated with an integer count
            n = int(self.n_max_frame / step_size + step_size)
            time_function = np.linspace(0, self.n_max_frame, n)
            return self._evaluate(time_function)
        tf = np.asarray(self.time_function, dtype=np.float64)
        canon = self._prim.canonical_grid if self._prim is not None else None
        if canon is not None and len(tf) == canon.size and np.array_equal(tf, self._prim_canonical_times()):
            return self._evaluate(tf, canon)
        return self._evaluate(tf)

    def _prim_canonical_times(self):
        if not hasattr(self._prim, "_canonical_times"):
            self._prim._canonical_times = self._prim.canonical_grid.tables()[2]
        return self._prim._canonical_times

    def evaluate(self, canonical_t):
        """Pose at arbitrary canonical time(s): (D,) for a scalar, (len(t), D) for an array
        (reference motion_spline.py:89-92)."""
        t = np.asarray(canonical_t, dtype=np.float64)
        out = self._evaluate(np.atleast_1d(t))
        return out[0] if t.ndim == 0 else out

    def get_buffered_motion_vector(self):
        if self.buffered_frames is None:
            self.buffered_frames = self.get_motion_vector()
        return self.buffered_frames

    def get_domain(self):
        return self.knots[0], self.knots[-1]
