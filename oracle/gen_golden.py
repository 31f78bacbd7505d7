#!/usr/bin/env python3
"""Generate tests/golden/*.npz by running the REFERENCE's own, unmodified
motion_primitive.py / motion_spline.py (imported from /root/reference through a
stub parent package that only supplies B_SPLINE_DEGREE = 3)
with the installed numpy/scipy/scikit-learn on seeded synthetic models.

Run in the build container only:   python oracle/gen_golden.py
The reference cannot travel to the GPU box; only the vectors written here do.
No reference source is copied: the fixtures hold inputs and expected outputs.
"""
import hashlib
import importlib
import os
import sys
import types
import warnings

import numpy as np

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
from morphablegraphs_amd import synthetic  # noqa: E402

REF_DIR = "/root/reference/morphablegraphs/motion_model"
OUT_DIR = os.environ.get("MG_GOLDEN_OUT") or os.path.join(ROOT, "tests", "golden")   # (MG_GOLDEN_OUT: oracle/check_golden.py regenerates beside, then compares)


def import_reference():
    warnings.filterwarnings("ignore")
    pkg = types.ModuleType("mg_ref_motion_model")
    pkg.__path__ = [REF_DIR]
    pkg.B_SPLINE_DEGREE = 3
    sys.modules["mg_ref_motion_model"] = pkg
    return importlib.import_module("mg_ref_motion_model.motion_primitive")


def model_digest(data):
    h = hashlib.sha256()
    for key in ("eigen_vectors_spatial", "mean_spatial_vector", "b_spline_knots_spatial",
                "gmm_weights", "gmm_means", "gmm_covars", "translation_maxima"):
        h.update(np.ascontiguousarray(np.asarray(data[key], dtype=np.float64)).tobytes())
    return h.hexdigest()


def run_case(ref, name, data, n_samples, eval_times, seed, store_model, n_score=64):
    mp = ref.MotionPrimitive(None)
    mp._initialize_from_json(data)
    np.random.seed(seed)
    S = mp.sample_low_dimensional_vector(n_samples)          # sklearn GaussianMixture.sample
    np.random.seed(seed + 1)
    X = mp.sample_low_dimensional_vector(n_score)
    X[: max(1, n_score // 8)] *= 3.0                          # push some rows into the tails
    coeffs = np.stack([mp.back_project_spatial_coeffs(s) for s in S])
    splines = [mp.back_project(s, use_time_parameters=False) for s in S]
    frames = np.stack([sp.get_motion_vector() for sp in splines])
    evals = np.stack([sp.evaluate(np.asarray(eval_times, dtype=float)) for sp in splines])
    evals_scalar = np.stack([splines[0].evaluate(float(t)) for t in eval_times])
    gmm = mp.gaussian_mixture_model
    logp = gmm.score_samples(X)
    logp_S = gmm.score_samples(S)
    out = dict(
        S=S, X=X, coeffs=coeffs, frames=frames, eval_times=np.asarray(eval_times, dtype=float),
        evals=evals, evals_scalar=evals_scalar, logp=logp, logp_S=logp_S,
        score_mean=np.float64(gmm.score(X)),
        precisions_cholesky=gmm.precisions_cholesky_,
        time_function=splines[0].time_function, knots=np.asarray(splines[0].knots),
        seed=np.int64(seed), digest=np.array(model_digest(data)),
        n_canonical_frames=np.int64(mp.get_n_canonical_frames()),
        n_spatial_components=np.int64(mp.get_n_spatial_components()),
    )
    if store_model:
        for key in ("eigen_vectors_spatial", "mean_spatial_vector", "b_spline_knots_spatial",
                    "gmm_weights", "gmm_means", "gmm_covars", "translation_maxima"):
            out["model_" + key] = np.asarray(data[key], dtype=np.float64)
        out["model_n_basis"] = np.int64(data["n_basis_spatial"])
        out["model_n_dim"] = np.int64(data["n_dim_spatial"])
    path = os.path.join(OUT_DIR, name + ".npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path) // 1024, "KiB")


def run_time_case(ref, name, data, n_samples, seed):
    """A model WITH the legacy time part: the mixture spans spatial + time latents (reference
    motion_model_constructor.py:424), back_project(s, False) reads s[:n_s], and
    _back_transform_gamma_to_canonical_time_function(s[n_s:]) runs (the inversion that follows it raises TypeError on
    every NumPy >= 1.18: SURVEY.md section 8c, so back_project(s, True) has no vector)."""
    mp = ref.MotionPrimitive(None)
    mp._initialize_from_json(data)
    assert mp.has_time_parameters
    np.random.seed(seed)
    S = mp.sample_low_dimensional_vector(n_samples)
    n_s = mp.get_n_spatial_components()
    splines = [mp.back_project(s, use_time_parameters=False) for s in S]
    frames = np.stack([sp.get_motion_vector() for sp in splines])
    ctf = np.stack([mp._back_transform_gamma_to_canonical_time_function(s[n_s:]) for s in S])
    gmm = mp.gaussian_mixture_model
    # The time-warped route, back_project(s, True): the reference hands np.linspace the FLOAT round(t(F-2)) * (1 / speed)
    # (motion_primitive.py:313-314), which NumPy >= 1.18 refuses; NumPy up to 1.17 truncated it.  The reference's own lines run
    # here unmodified with that behaviour of linspace restored for the duration of the calls.
    real_linspace = np.linspace
    np.linspace = lambda start, stop, num=50, **kw: real_linspace(start, stop, int(num), **kw)
    try:
        warped = {}
        for speed in (1.0, 1.6):
            stf = [mp._invert_canonical_to_sample_time_function(c, speed) for c in ctf]
            fr = [mp.back_project(s, use_time_parameters=True, speed=speed).get_motion_vector() for s in S]
            lens = np.array([len(t) for t in stf], dtype=np.int64)
            assert all(len(f) == n for f, n in zip(fr, lens))
            pad_t = np.full((len(S), lens.max()), np.nan)
            pad_f = np.full((len(S), lens.max(), frames.shape[2]), np.nan)
            for b in range(len(S)):
                pad_t[b, :lens[b]] = stf[b]
                pad_f[b, :lens[b]] = fr[b]
            tag = "speed%02d" % int(round(10 * speed))
            warped["sample_time_functions_" + tag] = pad_t
            warped["warped_frames_" + tag] = pad_f
            warped["warped_lengths_" + tag] = lens
    finally:
        np.linspace = real_linspace
    out = dict(S=S, frames=frames, canonical_time_functions=ctf, mean_temporal=mp._mean_temporal(), logp_S=gmm.score_samples(S),
               precisions_cholesky=gmm.precisions_cholesky_, seed=np.int64(seed), digest=np.array(model_digest(data)),
               n_canonical_frames=np.int64(mp.get_n_canonical_frames()), n_spatial_components=np.int64(n_s),
               n_time_components=np.int64(mp.get_n_time_components()),
               low_dimensional_parameters=np.asarray(splines[0].low_dimensional_parameters))
    for key in ("eigen_vectors_spatial", "mean_spatial_vector", "b_spline_knots_spatial", "gmm_weights", "gmm_means", "gmm_covars",
                "translation_maxima", "eigen_vectors_time", "mean_time_vector", "b_spline_knots_time"):
        out["model_" + key] = np.asarray(data[key], dtype=np.float64)
    out["model_n_basis"] = np.int64(data["n_basis_spatial"])
    out["model_n_dim"] = np.int64(data["n_dim_spatial"])
    out["model_n_basis_time"] = np.int64(data["n_basis_time"])
    out.update(warped)
    path = os.path.join(OUT_DIR, name + ".npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path) // 1024, "KiB")


# ---- the reference's optimiser wrappers on toy objectives (no model, no GPU): what HipLeastSquares / HipNumericalMinimizer
# must reproduce as wrappers -- the evaluation budget's meaning, the returned point -----------------------------------------
def toy_residuals(s, data=None):
    """6 residuals of 4 variables; rows of a batch are evaluated independently (the batched wrappers hand over (n, 4))."""
    s = np.asarray(s, dtype=np.float64)
    x = s[..., 0], s[..., 1], s[..., 2], s[..., 3]
    return np.stack([x[0] ** 2 + x[1] - 11.0, x[0] + x[1] ** 2 - 7.0, np.sin(x[2]) - 0.3, x[3] * x[0] - 1.0, 0.1 * (x[2] - x[3]), x[1] - 2.0 * x[3]], axis=-1)


def toy_scalar(s, data=None):
    r = toy_residuals(s)
    return np.sum(r * r, axis=-1)


def run_optimizer_drivers_case(name):
    """LeastSquares.run (least_squares.py:35-64) and NumericalMinimizer.run (numerical_minimizer.py:41-76), the reference's
    unmodified files imported through a stub parent package, on the toy objectives above."""
    pkg = types.ModuleType("mg_ref_optimization")
    pkg.__path__ = ["/root/reference/morphablegraphs/motion_generator/optimization"]
    sys.modules["mg_ref_optimization"] = pkg
    ls_mod = importlib.import_module("mg_ref_optimization.least_squares")
    nm_mod = importlib.import_module("mg_ref_optimization.numerical_minimizer")
    x0 = np.array([1.0, 1.5, 0.2, -0.4])
    out = {"x0": x0}
    for budget in (12, 30, 400):
        calls = [0]

        def counted(s, data):
            calls[0] += 1
            return toy_residuals(s, data)
        opt = ls_mod.LeastSquares({"max_iterations": budget, "verbose": False})
        opt.set_objective_function(counted)
        opt.set_objective_function_parameters(None)
        out["leastsq_%d" % budget] = np.asarray(opt.run(x0.copy()))
        out["leastsq_calls_%d" % budget] = np.int64(calls[0])
    for method, maxiter in (("BFGS", 6), ("BFGS", 200), ("L-BFGS-B", 50), ("Nelder-Mead", 40)):
        st = {"method": method, "max_iterations": maxiter, "diff_eps": 1e-6, "tolerance": 1e-10, "verbose": False}
        opt = nm_mod.NumericalMinimizer(st)
        opt.set_objective_function(lambda s, data: float(toy_scalar(s, data)))
        opt.set_objective_function_parameters(None)
        out["minimize_%s_%d" % (method.replace("-", "_"), maxiter)] = np.asarray(opt.run(x0.copy()))
    path = os.path.join(OUT_DIR, name + ".npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path) // 1024, "KiB")


def run_walk_32_case(ref, name, data, seed):
    """BASELINE configs[0]: one 'walk' primitive, 32 latent samples drawn the reference's way, back-projected by the reference
    (examples/run_construction.py:212-220 draws and back-projects samples of a freshly built model the same way); frames kept at
    40 of the 156 canonical frames (first, last two, 37 drawn) to hold the fixture to ~1 MB."""
    mp = ref.MotionPrimitive(None)
    mp._initialize_from_json(data)
    np.random.seed(seed)
    S = mp.sample_low_dimensional_vector(32)
    F = mp.get_n_canonical_frames()
    rows = np.unique(np.concatenate([[0, F - 2, F - 1], np.random.default_rng(seed).choice(F, 37, replace=False)]))
    frames = np.stack([mp.back_project(s, use_time_parameters=False).get_motion_vector()[rows] for s in S])
    gmm = mp.gaussian_mixture_model
    path = os.path.join(OUT_DIR, name + ".npz")
    np.savez_compressed(path, S=S, frame_rows=rows.astype(np.int64), frames_at_rows=frames, logp=gmm.score_samples(S), seed=np.int64(seed),
                        digest=np.array(model_digest(data)), n_canonical_frames=np.int64(F))
    print("wrote", path, os.path.getsize(path) // 1024, "KiB")


def run_trajectory_spline_case(name):
    """The spline under a TrajectoryConstraint (reference constraints/spatial_constraints/splines/{parameterized_spline,
    catmull_rom_spline,arc_length_map}.py, which need only numpy / scipy / matplotlib): points at given parameters and the
    arc length of the granularity-1000 table.  (The closest-point search: run_closest_point_case.)"""
    ps = import_reference_splines()
    rng = np.random.default_rng(77)
    out = {}
    for ci, n_points in enumerate((2, 5, 9)):
        cps = np.cumsum(np.column_stack([rng.uniform(20, 60, n_points), np.zeros(n_points), rng.uniform(-40, 40, n_points)]), axis=0)
        sp = ps.ParameterizedSpline(cps.tolist(), ps.SPLINE_TYPE_CATMULL_ROM)
        us = np.concatenate([[0.0, 1.0, 0.5, 1.0 / 3.0], rng.uniform(0, 1, 40)])
        out["control_points_%d" % ci] = cps
        out["parameters_%d" % ci] = us
        out["points_%d" % ci] = np.stack([np.asarray(sp.query_point_by_parameter(float(u)), dtype=np.float64) for u in us])
        out["full_arc_length_%d" % ci] = np.float64(sp.full_arc_length)
        # the arc-length parameterisation (query_point_by_absolute_arc_length: arc_length_map.py's table, searched and
        # interpolated; beyond the full length the last control point): what LocalTrajectoryConstraint, DiscreteTrajectoryConstraint
        # and TrajectorySetConstraint look their targets up with
        arcs = np.concatenate([[0.0, sp.full_arc_length, 0.5 * sp.full_arc_length, 1.25 * sp.full_arc_length], np.random.default_rng(780 + ci).uniform(0, sp.full_arc_length, 40)])   # (a stream of its own: the vectors above stay what they were)
        out["arc_lengths_%d" % ci] = arcs
        out["points_by_arc_%d" % ci] = np.stack([np.asarray(sp.query_point_by_absolute_arc_length(float(a)), dtype=np.float64) for a in arcs])
    out["n_cases"] = np.int64(3)
    path = os.path.join(OUT_DIR, name + ".npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path) // 1024, "KiB")


def import_reference_splines():
    """reference constraints/spatial_constraints/splines/*.py through a stub parent package (numpy / scipy / matplotlib only)"""
    if "mg_ref_splines" not in sys.modules:
        pkg = types.ModuleType("mg_ref_splines")
        pkg.__path__ = ["/root/reference/morphablegraphs/constraints/spatial_constraints/splines"]
        sys.modules["mg_ref_splines"] = pkg
    return importlib.import_module("mg_ref_splines.parameterized_spline")


class scalar_spline_parameter(object):
    """The ONE shim of the closest-point vectors, for the duration of the calls: scipy's L-BFGS-B hands the objective its
    parameter as a 1-element ndarray, and CatmullRomSpline.query_point_by_parameter (catmull_rom_spline.py:148-165) builds
    [u**3, u**2, u, 1] from it -- a ragged list that NumPy >= 1.24 refuses to turn into an array (older NumPy made an object
    array and the dot products went through).  The wrapper unwraps a size-1 array to the float it holds before the reference's
    own, unmodified method runs; nothing else is touched (the same kind of shim as np.linspace's int(num) for the time-warp
    vectors, run_time_case)."""

    def __enter__(self):
        self.cr = importlib.import_module("mg_ref_splines.catmull_rom_spline")
        self.orig = self.cr.CatmullRomSpline.query_point_by_parameter
        orig = self.orig

        def unwrapped(spline, u):
            if isinstance(u, np.ndarray) and u.size == 1:
                u = float(u.reshape(-1)[0])
            return orig(spline, u)
        self.cr.CatmullRomSpline.query_point_by_parameter = unwrapped
        return self

    def __exit__(self, *exc):
        self.cr.CatmullRomSpline.query_point_by_parameter = self.orig


def closest_point_tracks(rng, cps, walk_paths):
    """The joint tracks a case's spline is searched from: (name, track (T, 3), min_u at the first frame)"""
    cps = np.asarray(cps, dtype=np.float64)
    T = 156
    s = np.linspace(0.0, 1.0, T)
    chord = cps[0][None, :] + s[:, None] * (cps[-1] - cps[0])[None, :]
    span = float(np.linalg.norm(cps[-1] - cps[0]))
    wob = np.column_stack([0.03 * span * np.sin(7.0 * s), np.zeros(T), 0.04 * span * np.cos(5.0 * s)])
    tracks = [("near", chord + wob + rng.normal(0.0, 0.002 * span, (T, 3)) * np.array([1.0, 0.0, 1.0]), 0.0),
              # a hand: off the spline's plane, swinging in y, slower than the spline is long
              ("hand", cps[0][None, :] + 0.7 * s[:, None] * (cps[-1] - cps[0])[None, :] + np.column_stack(
                  [0.05 * span * np.sin(11.0 * s), 90.0 + 12.0 * np.sin(9.0 * s), 0.05 * span * np.sin(13.0 * s + 1.0)]), 0.0),
              # far from the spline (several spans to its side)
              ("far", chord + np.array([0.0, 0.0, 4.0 * span])[None, :] + wob, 0.0),
              # beyond the end: the track runs on for half a span past the last control point (the parameter meets its bound 1)
              ("beyond", cps[0][None, :] + 1.5 * s[:, None] * (cps[-1] - cps[0])[None, :] + 0.5 * wob, 0.0),
              # a later start on the spline, the track starting where the spline is at about a third
              ("late_start", cps[0][None, :] + (0.3 + 0.7 * s)[:, None] * (cps[-1] - cps[0])[None, :] + wob, 0.3),
              # a track that runs BACKWARDS along the spline: the bound (the previous frame's parameter) holds the point
              ("backwards", chord[::-1] + 0.5 * wob, 0.25)]
    for k, path in enumerate(walk_paths):
        tracks.append(("walk%d" % k, path, 0.0))
    return tracks


def run_closest_point_case(name):
    """ParameterizedSpline.find_closest_point_fast (splines/parameterized_spline.py:303-322: scipy L-BFGS-B on the distance over
    the spline parameter, bounds [min_u, 1], started at min_u), chained frame to frame exactly as
    TrajectoryConstraint.get_residual_vector does (trajectory_constraint.py:93-116: target, u = find_closest_point_fast(joint_position,
    min_u); errors[index] = norm(joint_position - target); min_u = u, starting from min_arc_length / full_arc_length) -- that class
    itself cannot be imported (its module pulls anim_utils through discrete_trajectory_constraint.py), so the six lines of its loop
    are restated here around the reference's own search.  Splines of 2, 5 and 9 control points; tracks near, far, beyond the
    end, with a late start, running backwards, a hand off the plane, and the root paths of the golden walks (walk_seed0) beside
    splines laid through them.  Per frame: parameter u, target point, distance."""
    ps = import_reference_splines()
    rng = np.random.default_rng(505)
    walk = np.load(os.path.join(OUT_DIR, "walk_seed0.npz"))
    walk_paths = [np.ascontiguousarray(walk["frames"][b][:, :3]) for b in range(2)]
    out = {"n_cases": np.int64(4)}
    with scalar_spline_parameter():
        for ci, n_points in enumerate((2, 5, 9, 7)):
            if ci < 3:
                cps = np.cumsum(np.column_stack([rng.uniform(20, 60, n_points), np.zeros(n_points), rng.uniform(-40, 40, n_points)]), axis=0)
                paths = []
            else:          # a spline laid through points of a golden walk's root path, displaced sideways: path following
                p0 = walk_paths[0]
                cps = p0[::26].copy()[:n_points]
                cps[:, 0] += np.linspace(0.0, 6.0, len(cps))
                cps[:, 2] -= np.linspace(0.0, 4.0, len(cps))
                paths = walk_paths
            sp = ps.ParameterizedSpline(cps.tolist(), ps.SPLINE_TYPE_CATMULL_ROM)
            out["control_points_%d" % ci] = cps
            out["full_arc_length_%d" % ci] = np.float64(sp.full_arc_length)
            tracks = closest_point_tracks(rng, cps, paths)
            out["track_names_%d" % ci] = np.array([t[0] for t in tracks])
            for ti, (tname, track, min_u0) in enumerate(tracks):
                track = np.ascontiguousarray(track, dtype=np.float64)
                us, targets, dists = np.empty(len(track)), np.empty((len(track), 3)), np.empty(len(track))
                min_u = min_u0                                    # trajectory_constraint.py:103 (min_arc_length / full_arc_length)
                for index, joint_position in enumerate(track):    # :104-113
                    target, u = sp.find_closest_point_fast(joint_position, min_u)
                    dists[index] = np.linalg.norm(joint_position - target)
                    us[index], targets[index] = float(np.ravel(u)[0]), target
                    min_u = u
                out["track_%d_%d" % (ci, ti)] = track
                out["min_u0_%d_%d" % (ci, ti)] = np.float64(min_u0)
                out["u_%d_%d" % (ci, ti)] = us
                out["target_%d_%d" % (ci, ti)] = targets
                out["distance_%d_%d" % (ci, ti)] = dists
                # single searches from the SAME lower bound in every frame (no chain): bound = the chain's previous parameter
                # is what the vectors above hold; here from a fixed bound, for the search alone
                fixed = np.array([float(np.ravel(sp.find_closest_point_fast(p, min_u0)[1])[0]) for p in track[::13]])
                out["u_fixed_bound_%d_%d" % (ci, ti)] = fixed
    path = os.path.join(OUT_DIR, name + ".npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path) // 1024, "KiB")


class _TrackNode(object):
    def __init__(self, track):
        self.track = track

    def get_global_position(self, frame, use_cache=False):
        return self.track[int(frame[0])].copy()


class _TrackSkeleton(object):
    """INPUT DATA in the shape the reference's classes ask their skeleton argument for: joint -> its global position per frame.
    A "frame" is the 1-element array [frame index]; nodes[joint].get_global_position(frame) returns the given position."""

    def __init__(self, tracks):
        self.nodes = {j: _TrackNode(np.asarray(t, dtype=np.float64)) for j, t in tracks.items()}

    def clear_cached_global_matrices(self):
        pass


class _IndexSpline(object):
    """aligned_spline for LocalTrajectoryConstraint.get_positions_from_spline: evaluate(idx) -> the frame token of frame idx"""

    def evaluate(self, idx):
        return np.array([float(idx)])


def run_per_frame_classes_case(name):
    """LocalTrajectoryConstraint (keyframe_constraints/local_trajectory_constraint.py:45-78) and TrajectorySetConstraint
    (trajectory_set_constraint.py:41-104), the reference's unmodified files imported through stub parent packages that supply only
    the label constants of spatial_constraints/__init__.py; their skeleton is an ARGUMENT: a duck-typed object that returns given
    joint tracks (input data, stored in the fixture).  Trajectories: the reference's ParameterizedSpline."""
    ps = import_reference_splines()
    _stub_spatial_packages()
    ltc = importlib.import_module("mg_ref_spatial.keyframe_constraints.local_trajectory_constraint")
    tsc = importlib.import_module("mg_ref_spatial.trajectory_set_constraint")
    rng = np.random.default_rng(606)
    T = 60
    s = np.linspace(0.0, 1.0, T)
    out = {}
    import contextlib
    import io
    for ci, n_points in enumerate((2, 5, 9)):
        cps = np.cumsum(np.column_stack([rng.uniform(20, 60, n_points), rng.uniform(-3, 3, n_points), rng.uniform(-40, 40, n_points)]), axis=0)
        cps2 = cps + np.array([5.0, 40.0, -8.0])
        traj, traj2 = ps.ParameterizedSpline(cps.tolist(), ps.SPLINE_TYPE_CATMULL_ROM), ps.ParameterizedSpline(cps2.tolist(), ps.SPLINE_TYPE_CATMULL_ROM)
        span = cps[-1] - cps[0]
        hips = cps[0][None, :] + (0.9 * s)[:, None] * span[None, :] + np.column_stack([2.0 * np.sin(9 * s), 0.5 * np.cos(4 * s), 3.0 * np.sin(6 * s + 0.5)])
        hand = hips + np.column_stack([4.0 * np.cos(12 * s), 40.0 + 6.0 * np.sin(10 * s), 5.0 * np.sin(8 * s)])
        sk = _TrackSkeleton({"Hips": hips, "LeftHand": hand})
        frames = [np.array([float(i)]) for i in range(T)]
        out["control_points_%d" % ci], out["control_points2_%d" % ci] = cps, cps2
        out["hips_%d" % ci], out["hand_%d" % ci] = hips, hand
        # LocalTrajectoryConstraint: arc length walked from start_t, squared xz distance per frame, summed
        for si, (start_t, nf) in enumerate(((0.0, T), (0.4 * traj.full_arc_length, 37))):
            desc = {"canonical_keyframe": 0, "semanticAnnotation": {"keyframeLabel": "none"}, "trajectory": traj, "start_t": start_t,
                    "n_canonical_frames": nf, "joint_name": "Hips"}
            c = ltc.LocalTrajectoryConstraint(sk, desc, 1.0, 1.0)
            out["local_start_t_%d_%d" % (ci, si)] = np.float64(start_t)
            out["local_n_frames_%d_%d" % (ci, si)] = np.int64(nf)
            out["local_residuals_%d_%d" % (ci, si)] = np.asarray(c.get_residual_vector_spline(_IndexSpline()), dtype=np.float64)
            out["local_error_%d_%d" % (ci, si)] = np.float64(c.evaluate_motion_spline(_IndexSpline()))
        # TrajectorySetConstraint: two joints, active ranges, start arc lengths
        full1, full2 = traj.full_arc_length, traj2.full_arc_length
        set_cases = [(((None, None), (None, None)), (0.0, 0.0)),                           # no active range: all residuals stay 0
                     (((0.0, 1.0e9), (None, None)), (3.0, 1.0)),                           # one trajectory always active
                     (((0.2 * full1, 0.7 * full1), (5.0, 0.5 * full2)), (3.0, 1.0))]       # ranges entered and left on the way
        for si, (ranges, arcs) in enumerate(set_cases):
            for t, r in zip((traj, traj2), ranges):
                t.range_start, t.range_end = r[0], r[1]
                t.is_active = (lambda tt: (lambda a: tt.range_start is not None and tt.range_start <= a <= tt.range_end))(t)   # trajectory_constraint.py:150-151
            c = tsc.TrajectorySetConstraint([traj, traj2], ["Hips", "LeftHand"], sk, 1.0, 1.0)
            c.set_number_of_canonical_frames(T)
            c.joint_arc_lengths = np.array(arcs, dtype=np.float64)
            with contextlib.redirect_stdout(io.StringIO()):
                res = np.asarray(c.get_residual_vector(frames), dtype=np.float64)
                err = c.evaluate_motion_sample(frames)
            out["set_ranges_%d_%d" % (ci, si)] = np.array([[np.nan if v is None else v for v in r] for r in ranges], dtype=np.float64)
            out["set_arc_lengths_%d_%d" % (ci, si)] = np.asarray(c.joint_arc_lengths, dtype=np.float64)
            out["set_residuals_%d_%d" % (ci, si)] = res
            out["set_error_%d_%d" % (ci, si)] = np.float64(err)
    out["n_cases"] = np.int64(3)
    path = os.path.join(OUT_DIR, name + ".npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path) // 1024, "KiB")


def _stub_spatial_packages():
    """Stub parents for the reference's constraints/spatial_constraints files: only the label constants of its __init__.py."""
    pkg = types.ModuleType("mg_ref_spatial")
    pkg.__path__ = ["/root/reference/morphablegraphs/constraints/spatial_constraints"]
    for label in ("TRAJECTORY", "KEYFRAME_POSITION", "KEYFRAME_DIR_2D", "KEYFRAME_POSE", "TWO_HAND_POSITION", "TRAJECTORY_SET", "KEYFRAME_LOOK_AT",
                  "KEYFRAME_FEET", "CA_CONSTRAINT", "KEYFRAME_RELATIVE_POSITION"):
        setattr(pkg, "SPATIAL_CONSTRAINT_TYPE_" + label, label.lower())
    pkg.__all__ = [k for k in vars(pkg) if k.startswith("SPATIAL_")]
    sys.modules["mg_ref_spatial"] = pkg
    kf = types.ModuleType("mg_ref_spatial.keyframe_constraints")
    kf.__path__ = ["/root/reference/morphablegraphs/constraints/spatial_constraints/keyframe_constraints"]
    sys.modules["mg_ref_spatial.keyframe_constraints"] = kf


class _FKNode(object):
    def __init__(self, orc, joints, animated, name):
        self.orc, self.joints, self.animated, self.name = orc, joints, animated, name

    def get_global_position(self, frame, use_cache=False):
        return self.orc.joint_global_position(np.asarray(frame, dtype=np.float64), self.joints, self.animated, self.name)


class _FKSkeleton(object):
    """The skeleton ARGUMENT of the keyframe classes: nodes[joint].get_global_position(frame) by the oracle's forward kinematics over
    synthetic.make_skeleton() (anim_utils, whose skeleton the reference would pass, is absent: the FK stays SELF-DEFINED; what this
    fixture pins is the classes' own arithmetic on the joint positions, which it also stores)."""

    def __init__(self, orc, joints, animated):
        self.nodes = {j[0]: _FKNode(orc, joints, animated, j[0]) for j in joints}


def run_keyframe_classes_case(ref, name):
    """TwoHandConstraintSet (keyframe_constraints/two_hand_constraint.py:33-93) and FeetConstraint (feet_constraint.py:30-55), the
    reference's unmodified files, on the reference's own MotionSpline objects (MotionPrimitive.back_project(s, False) of the walk
    model); evaluate_motion_spline / get_residual_vector_spline per candidate."""
    from oracle import mg_oracle as orc
    import contextlib
    import io
    _stub_spatial_packages()
    th = importlib.import_module("mg_ref_spatial.keyframe_constraints.two_hand_constraint")
    ft = importlib.import_module("mg_ref_spatial.keyframe_constraints.feet_constraint")
    data = synthetic.make_walk_primitive(seed=0)
    joints, animated = synthetic.make_skeleton()
    sk = _FKSkeleton(orc, joints, animated)
    mp = ref.MotionPrimitive(None)
    mp._initialize_from_json(data)
    np.random.seed(77)
    S = mp.sample_low_dimensional_vector(10)
    splines = [mp.back_project(s, use_time_parameters=False) for s in S]
    out = dict(S=S, digest=np.array(model_digest(data)))
    hands = [("LeftHand", "RightHand"), ("LeftHand_EndSite", "RightHand_EndSite")]
    two_hand = [(100, 0.5, [[30.0, 95.0, 10.0], [-20.0, 99.0, 14.0]], hands[0]),
                (0, 1.0, [[45.0, 120.0, -3.0], [-44.0, 118.0, 2.0]], hands[1]),
                (155, 2.0, [[10.0, 60.0, 30.0], [10.5, 60.0, 30.0]], hands[0])]            # targets half a unit apart
    for ci, (key, w, positions, names) in enumerate(two_hand):
        desc = {"canonical_keyframe": key, "semanticAnnotation": {"keyframeLabel": "none"}, "positions": [np.array(p) for p in positions],
                "orientations": [None, None], "joint": list(names), "n_canonical_frames": int(data["n_canonical_frames"])}
        c = th.TwoHandConstraintSet(sk, desc, 1.0, w)
        out["two_hand_keyframe_%d" % ci], out["two_hand_weight_%d" % ci] = np.int64(key), np.float64(w)
        out["two_hand_positions_%d" % ci], out["two_hand_joints_%d" % ci] = np.asarray(positions, dtype=np.float64), np.array(names)
        out["two_hand_residuals_%d" % ci] = np.array([c.get_residual_vector_spline(sp) for sp in splines], dtype=np.float64)
        out["two_hand_error_%d" % ci] = np.array([c.evaluate_motion_spline(sp) for sp in splines], dtype=np.float64)
        out["two_hand_hand_positions_%d" % ci] = np.array([c._get_global_hand_positions(sp.evaluate(key)) for sp in splines], dtype=np.float64)
        assert c.get_length_of_residual_vector() == 3
    feet = [(30, 1.0, [9.0, 3.0, 20.0], [-9.0, 2.0, 5.0]), (120, 1.5, [12.0, 0.0, 80.0], [-6.0, 10.0, 60.0])]
    for ci, (key, w, left, right) in enumerate(feet):
        desc = {"canonical_keyframe": key, "semanticAnnotation": {"keyframeLabel": "none"}, "left": np.array(left), "right": np.array(right)}
        c = ft.FeetConstraint(sk, desc, 1.0, w)
        with contextlib.redirect_stdout(io.StringIO()):
            out["feet_residuals_spline_%d" % ci] = np.array([c.get_residual_vector_spline(sp) for sp in splines], dtype=np.float64)   # ONE entry: left + right
            out["feet_residuals_%d" % ci] = np.array([c.get_residual_vector(sp.evaluate(key)) for sp in splines], dtype=np.float64)
            out["feet_error_%d" % ci] = np.array([c.evaluate_motion_spline(sp) for sp in splines], dtype=np.float64)
        out["feet_keyframe_%d" % ci], out["feet_weight_%d" % ci] = np.int64(key), np.float64(w)
        out["feet_left_%d" % ci], out["feet_right_%d" % ci] = np.asarray(left, dtype=np.float64), np.asarray(right, dtype=np.float64)
    out["n_two_hand"], out["n_feet"] = np.int64(len(two_hand)), np.int64(len(feet))
    path = os.path.join(OUT_DIR, name + ".npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path) // 1024, "KiB")


class _TimeNode(object):
    """graph.nodes[key] for TimeConstraints: back_project_time_function as the reference's wrapper forwards it on the legacy branch
    (motion_primitive_wrapper.py:233-237; the wrapper imports anim_utils, absent) -- the reference MotionPrimitive's own
    _back_transform_gamma_to_canonical_time_function on the time part of the vector."""

    def __init__(self, mp):
        self.mp = mp

    def back_project_time_function(self, s_vec):
        return self.mp._back_transform_gamma_to_canonical_time_function(np.asarray(s_vec)[self.mp.get_n_spatial_components():])

    def get_gaussian_mixture_model(self):
        # time_constraints.py:95 indexes the return value of score(X): sklearn's old GMM.score returned the per-sample
        # log-likelihoods, GaussianMixture.score returns their mean as a scalar (IndexError on every sklearn >= 0.20).  The
        # reference's line runs here unmodified over the mixture's per-sample scores.
        return _Bag(score=self.mp.gaussian_mixture_model.score_samples)


class _Bag(object):
    def __init__(self, **kw):
        self.__dict__.update(kw)


def run_time_constraints_case(ref, name, data):
    """TimeConstraints (constraints/time_constraints.py:25-110), the reference's unmodified file, over a three-step graph walk on
    the time_model fixture's primitive; the walk's steps take that fixture's sampled vectors S[0..2] as parameters."""
    import contextlib
    import io
    pkg = types.ModuleType("mg_ref_constraints")
    pkg.__path__ = ["/root/reference/morphablegraphs/constraints"]
    sys.modules["mg_ref_constraints"] = pkg
    tc_mod = importlib.import_module("mg_ref_constraints.time_constraints")
    mp = ref.MotionPrimitive(None)
    mp._initialize_from_json(data)
    n_s, n_t, F = mp.get_n_spatial_components(), mp.get_n_time_components(), mp.get_n_canonical_frames()
    np.random.seed(41)                                                   # run_time_case's seed: the same S
    base = mp.sample_low_dimensional_vector(9)[:3]
    steps = [_Bag(node_key="tm", parameters=np.array(b), n_spatial_components=n_s, n_time_components=n_t) for b in base]
    walk = _Bag(steps=steps)
    graph = _Bag(nodes={"tm": _TimeNode(mp)}, skeleton=_Bag(frame_time=0.02))
    rng = np.random.default_rng(4)
    out = dict(base=base, frame_time=np.float64(0.02), digest=np.array(model_digest(data)))
    cases = [(1, 3, [(0, F // 2, 1.1), (1, F - 1, 3.9), (2, 5, 4.2), (5, 1, 1.0), (1, F + 3, 2.0)]),
             (0, 3, [(0, 0, 0.3), (2, F - 1, 3.0)]),
             (2, 3, [(0, 17, 2.9)])]
    for ci, (start, end, constraint_list) in enumerate(cases):
        tc = tc_mod.TimeConstraints(graph, walk, start, end, constraint_list)
        S = 0.5 * rng.standard_normal((6, (end - start) * n_t))
        with contextlib.redirect_stdout(io.StringIO()):
            err = np.array([tc.evaluate_graph_walk(s, graph, walk) for s in S], dtype=np.float64)
        ll = np.array([tc.get_average_loglikelihood(s, graph, walk) for s in S], dtype=np.float64)
        out["start_step_%d" % ci], out["end_step_%d" % ci] = np.int64(start), np.int64(end)
        out["constraint_list_%d" % ci] = np.asarray(constraint_list, dtype=np.float64)
        out["S_%d" % ci], out["error_%d" % ci], out["loglikelihood_%d" % ci] = S, err, ll
        out["start_keyframe_%d" % ci] = np.float64(tc.start_keyframe)
        out["initial_guess_%d" % ci] = np.asarray(tc.get_initial_guess(walk), dtype=np.float64)
    out["n_cases"] = np.int64(len(cases))
    path = os.path.join(OUT_DIR, name + ".npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path) // 1024, "KiB")


def main():
    ref = import_reference()
    os.makedirs(OUT_DIR, exist_ok=True)
    if len(sys.argv) > 1 and sys.argv[1] == "time_model":   # only this fixture (the archives carry time stamps: untouched ones stay byte-identical)
        timed = synthetic.make_primitive(seed=13, n_components=12, n_frames=60, n_dim=15, n_gmm=3, name="timed",
                                         n_time_components=3, n_basis_time=8)
        run_time_case(ref, "time_model", timed, 9, 41)
        return
    if len(sys.argv) > 1 and sys.argv[1] == "round5b":      # round 5, second batch: the keyframe classes and the time constraints
        run_keyframe_classes_case(ref, "keyframe_classes")
        timed = synthetic.make_primitive(seed=13, n_components=12, n_frames=60, n_dim=15, n_gmm=3, name="timed",
                                         n_time_components=3, n_basis_time=8)
        run_time_constraints_case(ref, "time_constraints", timed)
        return
    if len(sys.argv) > 1 and sys.argv[1] == "round5":       # the fixtures added in round 5
        run_closest_point_case("trajectory_closest_point")
        run_per_frame_classes_case("per_frame_classes")
        return
    if len(sys.argv) > 1 and sys.argv[1] == "round4":       # the fixtures added in round 4
        run_optimizer_drivers_case("optimizer_drivers")
        run_walk_32_case(ref, "walk_32", synthetic.make_walk_primitive(seed=0), 17)
        return
    # (i) tiny model, non-unit translation maxima, times incl. out-of-range (extrapolated) ones
    tiny = synthetic.make_tiny_primitive(seed=1, translation_maxima=(1.5, 2.0, 0.5))
    run_case(ref, "tiny_tm", tiny, 5, [0.0, 0.25, 5.5, 10.999, 11.0, 11.5, 12.0, -0.5], 3, True, n_score=16)
    # (ii) walk-sized, realistic magnitudes, v3 load path translation maxima [1,1,1]
    walk = synthetic.make_walk_primitive(seed=0)
    run_case(ref, "walk_seed0", walk, 4, [0.0, 77.5, 155.0, 156.0, 33.3, 154.999], 0, False)
    # (iii) walk-sized, scaled translation, Dirichlet weights, unrealistic O(1) magnitudes
    walk_tm = synthetic.make_walk_primitive(seed=7, translation_maxima=(1.5, 2.0, 0.5),
                                            dirichlet_weights=True, realistic=False)
    run_case(ref, "walk_seed7_tm", walk_tm, 3, [0.0, 77.5, 155.0], 11, False)
    # (iv) single component with a nearly singular covariance
    k1 = synthetic.make_primitive(seed=5, n_components=6, n_frames=40, n_dim=11, n_gmm=1, name="k1")
    rng = np.random.default_rng(55)
    a = rng.standard_normal((6, 2))
    k1["gmm_covars"] = [(a @ a.T + 1e-6 * np.eye(6)).tolist()]
    run_case(ref, "k1_near_singular", k1, 4, [0.0, 19.5, 39.0, 40.0], 21, True, n_score=16)
    # (v) small odd shape (D not a multiple of 4, L not a multiple of 4)
    odd = synthetic.make_primitive(seed=9, n_components=13, n_frames=47, n_dim=15, n_gmm=3, name="odd",
                                   translation_maxima=(2.0, 1.0, 3.0))
    run_case(ref, "odd_shape", odd, 6, [0.0, 23.0, 46.0, 47.0], 31, True, n_score=32)
    # (vi) a model with the legacy time part: mixture over 12 spatial + 3 time latents
    timed = synthetic.make_primitive(seed=13, n_components=12, n_frames=60, n_dim=15, n_gmm=3, name="timed",
                                     n_time_components=3, n_basis_time=8)
    run_time_case(ref, "time_model", timed, 9, 41)
    run_trajectory_spline_case("trajectory_spline")
    run_optimizer_drivers_case("optimizer_drivers")
    run_walk_32_case(ref, "walk_32", walk, 17)
    run_closest_point_case("trajectory_closest_point")
    run_per_frame_classes_case("per_frame_classes")


if __name__ == "__main__":
    main()
