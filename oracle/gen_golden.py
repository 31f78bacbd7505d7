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
OUT_DIR = os.path.join(ROOT, "tests", "golden")


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
    arc length of the granularity-1000 table.  (find_closest_point_fast itself raises ValueError under the installed NumPy:
    the closest-point search has no vector.)"""
    pkg = types.ModuleType("mg_ref_splines")
    pkg.__path__ = ["/root/reference/morphablegraphs/constraints/spatial_constraints/splines"]
    sys.modules["mg_ref_splines"] = pkg
    ps = importlib.import_module("mg_ref_splines.parameterized_spline")
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


def main():
    ref = import_reference()
    os.makedirs(OUT_DIR, exist_ok=True)
    if len(sys.argv) > 1 and sys.argv[1] == "time_model":   # only this fixture (the archives carry time stamps: untouched ones stay byte-identical)
        timed = synthetic.make_primitive(seed=13, n_components=12, n_frames=60, n_dim=15, n_gmm=3, name="timed",
                                         n_time_components=3, n_basis_time=8)
        run_time_case(ref, "time_model", timed, 9, 41)
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


if __name__ == "__main__":
    main()
