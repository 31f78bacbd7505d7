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
