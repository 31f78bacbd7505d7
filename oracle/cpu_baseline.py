"""ORACLE-side CPU baselines for bench.py's `cpu_baseline` leg (test/measurement
infrastructure, never the product path).

reference_shaped_rate: the reference's own per-candidate call pattern and libraries,
restated without its source -- E @ alpha + mean -> reshape -> root rescale
(reference motion_primitive.py:236-256), 79 x scipy.interpolate.splev over
np.linspace(0, F, F) (reference motion_spline.py:71-86, motion_primitive.py:233) and
sklearn GaussianMixture.score_samples on one row (reference objective_functions.py:256
call pattern) -- single process, single core.
c_port_rate: the plain-C float64 oracle (oracle/mg_oracle.c) on one core.
"""
import time

import numpy as np


def reference_shaped_rate(data, S, budget_s=12.0, max_candidates=8192):
    import scipy.interpolate as si
    from sklearn.mixture import GaussianMixture
    from sklearn.mixture._gaussian_mixture import _compute_precision_cholesky
    E = np.transpose(np.array(data["eigen_vectors_spatial"]))
    mean = np.array(data["mean_spatial_vector"])
    tm = np.array(data["translation_maxima"])
    nb, nd = int(data["n_basis_spatial"]), int(data["n_dim_spatial"])
    knots = np.asarray(data["b_spline_knots_spatial"])
    F = int(data["n_canonical_frames"])
    gmm = GaussianMixture(n_components=len(data["gmm_weights"]), covariance_type="full")
    gmm.weights_ = np.array(data["gmm_weights"])
    gmm.means_ = np.array(data["gmm_means"])
    gmm.covariances_ = np.array(data["gmm_covars"])
    gmm.precisions_cholesky_ = _compute_precision_cholesky(gmm.covariances_, "full")
    n = 0
    t0 = time.perf_counter()
    checksum = 0.0
    while n < max_candidates:
        s = np.asarray(S[n % len(S)], dtype=np.float64)
        coefs = np.dot(E, s)
        coefs += mean
        coefs = coefs.reshape((nb, nd))
        coefs[:, :3] *= tm
        tf = np.linspace(0, F, int(F * 1.0))
        ct = coefs.T
        frames = np.asarray([si.splev(tf, (knots, ct[i], 3)) for i in range(nd)]).T
        lp = gmm.score_samples(s.reshape(1, -1))
        checksum += float(frames[-1, 0]) + float(lp[0])
        n += 1
        if time.perf_counter() - t0 > budget_s:
            break
    dt = time.perf_counter() - t0
    return {"rate": n / dt, "n": n, "seconds": dt, "checksum": checksum}


def c_port_rate(data, S, budget_s=6.0):
    from oracle import c_oracle
    cp = c_oracle.COraclePrimitive(data)
    S = np.ascontiguousarray(S, dtype=np.float64)
    n, chunk = 0, 64
    t0 = time.perf_counter()
    while True:
        part = S[(n % len(S)):(n % len(S)) + chunk]
        cp.frames_f64(part)
        cp.log_prob_f64(part)
        n += len(part)
        if time.perf_counter() - t0 > budget_s:
            break
    dt = time.perf_counter() - t0
    return {"rate": n / dt, "n": n, "seconds": dt}


def vectorised_rate(data, S, budget_s=5.0, block=2048):
    """Best-effort vectorised CPU form of the same step (SURVEY 8(d) baseline (ii)), all BLAS threads the host gives:
    one GEMM for the coefficients (S E'^T + mean'), one GEMM with the dense (F x n_basis) basis matrix for the
    frames, float32 frames materialised like the GPU path, and sklearn's batched score_samples for log p(x)."""
    from sklearn.mixture import GaussianMixture
    from sklearn.mixture._gaussian_mixture import _compute_precision_cholesky
    from oracle import mg_oracle as orc
    nb, nd = int(data["n_basis_spatial"]), int(data["n_dim_spatial"])
    F = int(data["n_canonical_frames"])
    tm = np.array(data["translation_maxima"], dtype=np.float64)
    scale = np.ones(nb * nd)
    for d in range(3):
        scale[d::nd] = tm[d]
    Et = (np.array(data["eigen_vectors_spatial"], dtype=np.float64) * scale[None, :]).astype(np.float32)   # (L, nb*nd)
    mean = (np.array(data["mean_spatial_vector"], dtype=np.float64) * scale).astype(np.float32)
    knots = np.asarray(data["b_spline_knots_spatial"], dtype=np.float64)
    i0, w = orc.basis_rows(knots, np.linspace(0, F, F))
    Bm = np.zeros((F, nb), dtype=np.float32)
    for f in range(F):
        Bm[f, i0[f]:i0[f] + 4] = w[f]
    gmm = GaussianMixture(n_components=len(data["gmm_weights"]), covariance_type="full")
    gmm.weights_ = np.array(data["gmm_weights"])
    gmm.means_ = np.array(data["gmm_means"])
    gmm.covariances_ = np.array(data["gmm_covars"])
    gmm.precisions_cholesky_ = _compute_precision_cholesky(gmm.covariances_, "full")
    S32 = np.ascontiguousarray(S, dtype=np.float32)
    threads = 1
    try:
        from threadpoolctl import threadpool_info
        threads = max([int(p.get("num_threads", 1)) for p in threadpool_info() if p.get("user_api") == "blas"] or [1])
    except Exception:
        pass
    frames = np.empty((block, F, nd), dtype=np.float32)
    for _ in range(2):   # BLAS thread pools and page faults of the output out of the timed region
        np.matmul(Bm, (S32[:block] @ Et + mean).reshape(-1, nb, nd), out=frames[:len(S32[:block])])
    n = 0
    t0 = time.perf_counter()
    checksum = 0.0
    while True:
        part = S32[(n % len(S32)):(n % len(S32)) + block]
        coeffs = (part @ Et + mean).reshape(len(part), nb, nd)            # (b, nb, nd)
        np.matmul(Bm, coeffs, out=frames[:len(part)])                      # (b, F, nd) float32, materialised
        lp = gmm.score_samples(part.astype(np.float64))
        checksum += float(frames[-1, -1, 0]) + float(lp[0])
        n += len(part)
        if time.perf_counter() - t0 > budget_s:
            break
    dt = time.perf_counter() - t0
    return {"rate": n / dt, "n": n, "seconds": dt, "checksum": checksum, "threads": threads}
