"""The target spline of the trajectory constraints on the host: a Catmull-Rom curve through control points and its arc-length
parameterisation (reference constraints/spatial_constraints/splines/catmull_rom_spline.py:66-168, parameterized_spline.py:131-155,
arc_length_map.py:45-120), vectorised over query values.  The device holds the same curve as per-segment cubics
(mg_trajectory_create); this is its host twin for the constraints that look targets up BY ARC LENGTH (LocalTrajectoryConstraint,
DiscreteTrajectoryConstraint, TrajectorySetConstraint).  Pinned by tests/golden/trajectory_spline.npz (points by parameter, full
arc length and points by absolute arc length made by the reference's own code)."""
import numpy as np

_BASE = np.array([[-1.0, 3.0, -3.0, 1.0], [2.0, -5.0, 4.0, -1.0], [-1.0, 0.0, 1.0, 0.0], [0.0, 2.0, 0.0, 0.0]])


class CatmullRomPath(object):
    def __init__(self, control_points, granularity=1000):
        P = np.asarray(control_points, dtype=np.float64)
        if P.ndim != 2 or len(P) < 2:
            raise ValueError("a trajectory needs at least two control points")
        self.control_points = P
        self.n_seg = len(P) - 1
        padded = np.vstack([P[:1], P, P[-1:], P[-1:]])                       # catmull_rom_spline.py:66-71
        # segment s (0-based) uses padded[s .. s + 3]; point = 0.5 * [t^3 t^2 t 1] . base . ctrl
        self.poly = np.stack([0.5 * (_BASE @ padded[s:s + 4]) for s in range(self.n_seg)])      # (n_seg, 4, dims)
        self.granularity = int(granularity)
        us = np.arange(self.granularity + 1) / float(self.granularity)
        pts = self.point(us)
        steps = np.linalg.norm(pts[1:] - pts[:-1], axis=1)
        acc = np.concatenate([[0.0], np.cumsum(steps)])
        self.full_arc_length = float(acc[-1])
        if self.full_arc_length == 0.0:
            raise ValueError("Not enough control points in trajectory constraint definition")
        self._table_u, self._table_rel = us, acc / self.full_arc_length       # arc_length_map.py:45-71

    def point(self, u):
        """query_point_by_parameter for an array of parameters -> (n, dims)"""
        u = np.atleast_1d(np.asarray(u, dtype=np.float64))
        scaled = self.n_seg * u
        index = np.minimum(np.floor(scaled).astype(np.int64), self.n_seg)
        t = scaled - index
        inside = index < self.n_seg
        seg = np.where(inside, index, 0)
        A = self.poly[seg]                                                    # (n, 4, dims)
        out = ((A[:, 0] * t[:, None] + A[:, 1]) * t[:, None] + A[:, 2]) * t[:, None] + A[:, 3]
        out[~inside] = self.control_points[-1]
        return out

    def parameter_of_relative_arc_length(self, rel):
        """map_relative_arc_length_to_parameter (arc_length_map.py:97-110): the table searched for the bounding entries and
        interpolated linearly; below / above the table its first / last parameter"""
        return np.interp(np.asarray(rel, dtype=np.float64), self._table_rel, self._table_u)

    def point_by_absolute_arc_length(self, arc):
        """query_point_by_absolute_arc_length (parameterized_spline.py:131-148) for an array: beyond the full arc length the
        last control point"""
        arc = np.atleast_1d(np.asarray(arc, dtype=np.float64))
        out = self.point(self.parameter_of_relative_arc_length(np.minimum(arc, self.full_arc_length) / self.full_arc_length))
        out[arc > self.full_arc_length] = self.control_points[-1]
        return out
