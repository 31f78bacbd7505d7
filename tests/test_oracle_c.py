"""The plain-C oracle (oracle/mg_oracle.c) against the NumPy oracle and the
golden vectors made by the reference's own code.  CPU only."""
import numpy as np

from oracle import c_oracle
from oracle import mg_oracle as orc


def test_c_f64_frames_match_golden(golden_case):
    name, data, g = golden_case
    cp = c_oracle.COraclePrimitive(data)
    scale = max(1.0, np.abs(g["frames"]).max())
    np.testing.assert_allclose(cp.frames_f64(g["S"]), g["frames"], rtol=0, atol=2e-12 * scale)
    np.testing.assert_allclose(cp.coeffs_f64(g["S"]), g["coeffs"], rtol=0, atol=2e-12 * scale)
    np.testing.assert_allclose(cp.frames_f64(g["S"], g["eval_times"]), g["evals"], rtol=0, atol=2e-12 * scale)
    np.testing.assert_array_equal(cp.canonical_time_function(), g["time_function"])


def test_c_basis_rows_equal_numpy_oracle(golden_case):
    name, data, g = golden_case
    cp = c_oracle.COraclePrimitive(data)
    tp = np.concatenate([g["time_function"], g["eval_times"]])
    i0_c, w_c = cp.basis_rows(tp)
    i0_p, w_p = orc.basis_rows(cp.knots, tp)
    np.testing.assert_array_equal(i0_c, i0_p)
    np.testing.assert_array_equal(w_c, w_p)          # same operation order -> same bits
    assert np.all(i0_c >= 0) and np.all(i0_c + 3 < cp.NB)


def test_c_gmm_matches_golden(golden_case):
    name, data, g = golden_case
    cp = c_oracle.COraclePrimitive(data)
    pc = g["precisions_cholesky"]
    np.testing.assert_allclose(cp.prec_chol, pc, rtol=1e-9, atol=1e-9 * np.abs(pc).max())
    np.testing.assert_allclose(cp.log_prob_f64(g["X"]), g["logp"], rtol=1e-9, atol=1e-7)


def test_c_f32_model_within_tolerance_of_reference(golden_case):
    """The float32 arithmetic contract the HIP kernels implement stays within the
    north-star tolerance of the reference's float64 frames:
    |err| <= 1e-5 + 2^-23 |ref|  (1e-5 abs, plus half an f32 ulp of the value itself
    for root-translation magnitudes where 1e-5 is below f32 resolution)."""
    name, data, g = golden_case
    cp = c_oracle.COraclePrimitive(data)
    got = cp.frames_f32model(g["S"]).astype(np.float64)
    ref = g["frames"]
    tol = 1e-5 + 2.0 ** -23 * np.abs(ref)
    assert np.all(np.abs(got - ref) <= tol), float(np.max(np.abs(got - ref) / tol))


def test_c_argmin_rule():
    assert c_oracle.first_min_argmin(np.array([3.0, 1.0, 1.0, 2.0])) == (1, 1.0)
    assert c_oracle.first_min_argmin(np.array([np.nan, 2.0, np.nan, 2.0], dtype=np.float32)) == (1, 2.0)
    assert c_oracle.first_min_argmin(np.array([], dtype=np.float64))[0] == 0


def test_c_keyframe_errors_equal_numpy_oracle():
    from morphablegraphs_amd import synthetic
    data = synthetic.make_primitive(seed=4, n_components=8, n_frames=30, n_dim=11, n_gmm=2)
    cp = c_oracle.COraclePrimitive(data)
    po = orc.OraclePrimitive(data)
    rng = np.random.default_rng(0)
    S = rng.standard_normal((7, 8))
    cons_py = [
        {"type": "position", "t": 29.0, "weight": 1.0, "target": [10.0, None, -20.0]},
        {"type": "position", "t": 14.5, "weight": 0.5, "target": [1.0, 2.0, 3.0]},
        {"type": "direction", "t": 29.0, "weight": 2.0, "target": [0.3, -1.0]},
    ]
    nan = np.nan
    cons_c = np.array([[0, 29.0, 1.0, 10.0, nan, -20.0, 0, 0],
                       [0, 14.5, 0.5, 1.0, 2.0, 3.0, 0, 0],
                       [1, 29.0, 2.0, 0.3, -1.0, 0.0, 0.0, 1.0]])
    # type-1 rows carry the reference direction in columns 5..7
    cons_c[2, 5:8] = [0.0, 0.0, 1.0]
    np.testing.assert_allclose(cp.keyframe_errors_f64(S, cons_c), po.keyframe_errors(S, cons_py), rtol=1e-10, atol=1e-10)


def test_cpu_baselines_compute_the_same_step():
    """bench.py's cpu_baseline legs are valid computations of the step, not strawmen: the reference-shaped loop, the
    C port and the vectorised GEMM form run on a small model and a short budget, and the vectorised form's frames /
    log p(x) agree with the float64 oracle to float32 accuracy."""
    import numpy as np
    from morphablegraphs_amd import synthetic
    from oracle import cpu_baseline, c_oracle, mg_oracle as orc
    data = synthetic.make_walk_primitive(seed=0)
    S = np.random.default_rng(0).standard_normal((64, 40)).astype(np.float32)
    for fn in (cpu_baseline.reference_shaped_rate, cpu_baseline.c_port_rate, cpu_baseline.vectorised_rate):
        r = fn(data, S, budget_s=0.3)
        assert r["n"] > 0 and r["rate"] > 0 and r["seconds"] > 0
    # the vectorised form, restated on 8 rows, against the oracle
    op = orc.OraclePrimitive(data)
    nb, nd, F = 31, 79, 156
    Et = np.array(data["eigen_vectors_spatial"], dtype=np.float64)
    mean = np.array(data["mean_spatial_vector"], dtype=np.float64)
    tm = np.array(data["translation_maxima"], dtype=np.float64)
    scale = np.ones(nb * nd)
    for d in range(3):
        scale[d::nd] = tm[d]
    i0, w = orc.basis_rows(op.knots, np.linspace(0, F, F))
    Bm = np.zeros((F, nb), dtype=np.float32)
    for f in range(F):
        Bm[f, i0[f]:i0[f] + 4] = w[f]
    coeffs = (S[:8] @ (Et * scale).astype(np.float32) + (mean * scale).astype(np.float32)).reshape(8, nb, nd)
    frames = np.matmul(Bm, coeffs)
    ref = op.back_project_frames_batch(S[:8].astype(np.float64))
    np.testing.assert_allclose(frames, ref, rtol=0, atol=2e-4 * max(1.0, np.abs(ref).max()))
