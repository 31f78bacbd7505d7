"""ORACLE (test infrastructure) -- ctypes loader for oracle/libmg_oracle.so
(built from oracle/mg_oracle.c by oracle/Makefile).  Importable only from
tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg."""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libmg_oracle.so")
_lib = None

_dp = C.POINTER(C.c_double)
_fp = C.POINTER(C.c_float)
_ip = C.POINTER(C.c_int32)


def build(force=False):
    src = os.path.join(HERE, "mg_oracle.c")
    if force or not os.path.exists(LIB_PATH) or os.path.getmtime(LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-s", "-C", HERE, "libmg_oracle.so"])
    return LIB_PATH


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            build()
        _lib = C.CDLL(LIB_PATH)
        _lib.orc_first_min_argmin_f64.restype = C.c_int64
        _lib.orc_first_min_argmin_f32.restype = C.c_int64
        _lib.orc_precision_cholesky.restype = C.c_int
        _lib.orc_root_split_estimate.restype = C.c_double
    return _lib


def _d(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _ptr(a, t):
    return a.ctypes.data_as(t)


class COraclePrimitive(object):
    """Same model dict as the reference's MotionPrimitive._initialize_from_json."""

    def __init__(self, data):
        self.E = _d(np.transpose(np.asarray(data["eigen_vectors_spatial"], dtype=np.float64)))  # (NB*D, L)
        self.mean = _d(data["mean_spatial_vector"])
        self.tm = _d(data["translation_maxima"])
        self.NB = int(data["n_basis_spatial"])
        self.D = int(data["n_dim_spatial"])
        self.L = self.E.shape[1]
        self.F = int(data["n_canonical_frames"])
        self.knots = _d(data["b_spline_knots_spatial"])
        self.weights = _d(data["gmm_weights"])
        self.means = _d(data["gmm_means"])
        self.covars = _d(data["gmm_covars"])
        self.K = len(self.weights)
        self.prec_chol = np.empty_like(self.covars)
        rc = lib().orc_precision_cholesky(_ptr(self.covars, _dp), self.K, self.L, _ptr(self.prec_chol, _dp))
        if rc != 0:
            raise ValueError("covariance %d not positive definite" % (rc - 1))
        # the root channels' mode of the float32 contract (include/mg_hip.h, mg_primitive_root_mode): the mean/delta split
        # where its error estimate is at most 5e-6, the float64 pipeline otherwise
        Lg = self.means.shape[1] if self.K else self.L
        self.root_split_estimate = float(lib().orc_root_split_estimate(
            _ptr(self.E, _dp), _ptr(self.tm, _dp), self.NB, self.D, self.L, self.K, Lg,
            _ptr(self.weights, _dp), _ptr(self.means, _dp), _ptr(self.covars, _dp)))
        self.root_split = bool(np.isfinite(self.root_split_estimate) and self.root_split_estimate <= 5e-6)

    def canonical_time_function(self):
        out = np.empty(self.F)
        lib().orc_canonical_time_function(self.F, _ptr(out, _dp))
        return out

    def basis_rows(self, tp):
        tp = _d(np.atleast_1d(tp))
        i0 = np.empty(len(tp), dtype=np.int32)
        w = np.empty((len(tp), 4))
        lib().orc_basis_rows(_ptr(self.knots, _dp), len(self.knots), _ptr(tp, _dp), len(tp), _ptr(i0, _ip), _ptr(w, _dp))
        return i0, w

    def frames_f64(self, S, tp=None):
        S = _d(np.atleast_2d(S))
        tp = self.canonical_time_function() if tp is None else _d(np.atleast_1d(tp))
        out = np.empty((S.shape[0], len(tp), self.D))
        lib().orc_back_project_frames_f64(_ptr(self.E, _dp), _ptr(self.mean, _dp), _ptr(self.tm, _dp),
                                          self.NB, self.D, self.L, _ptr(self.knots, _dp), _ptr(tp, _dp), len(tp),
                                          _ptr(S, _dp), C.c_int64(S.shape[0]), C.c_int64(S.shape[1]), _ptr(out, _dp))
        return out

    def coeffs_f64(self, S):
        S = _d(np.atleast_2d(S))
        out = np.empty((S.shape[0], self.NB, self.D))
        lib().orc_back_project_coeffs_f64(_ptr(self.E, _dp), _ptr(self.mean, _dp), _ptr(self.tm, _dp),
                                          self.NB, self.D, self.L, _ptr(S, _dp), C.c_int64(S.shape[0]),
                                          C.c_int64(S.shape[1]), _ptr(out, _dp))
        return out

    def frames_f32model(self, S, tp=None, root_split=None):
        """root_split: None / False = the float64 pipeline for the root channels (the library's default), True = the mean/delta
        split (MG_OPT_ROOT_MODE 2, or 3 where self.root_split -- the contract's accuracy gate -- allows it)."""
        S = _d(np.atleast_2d(S))   # float32 callers pass exactly representable values
        tp = self.canonical_time_function() if tp is None else _d(np.atleast_1d(tp))
        out = np.empty((S.shape[0], len(tp), self.D), dtype=np.float32)
        split = bool(root_split)
        lib().orc_back_project_frames_f32model_mode(_ptr(self.E, _dp), _ptr(self.mean, _dp), _ptr(self.tm, _dp),
                                                    self.NB, self.D, self.L, _ptr(self.knots, _dp), _ptr(tp, _dp), len(tp),
                                                    _ptr(S, _dp), C.c_int64(S.shape[0]), C.c_int64(S.shape[1]),
                                                    C.c_int(1 if split else 0), _ptr(out, _fp))
        return out

    def log_prob_f64(self, X):
        X = _d(np.atleast_2d(X))
        out = np.empty(X.shape[0])
        lib().orc_gmm_log_prob_f64(_ptr(self.weights, _dp), _ptr(self.means, _dp), _ptr(self.prec_chol, _dp),
                                   self.K, self.L, _ptr(X, _dp), C.c_int64(X.shape[0]), C.c_int64(X.shape[1]),
                                   _ptr(out, _dp))
        return out

    def keyframe_errors_f64(self, S, cons):
        """cons: (n, 8) float64 rows {type, t, weight, a, b, c|rx, ry, rz} (see mg_oracle.c)."""
        S = _d(np.atleast_2d(S))
        cons = _d(np.atleast_2d(cons))
        out = np.empty(S.shape[0])
        lib().orc_keyframe_errors_f64(_ptr(self.E, _dp), _ptr(self.mean, _dp), _ptr(self.tm, _dp),
                                      self.NB, self.D, self.L, _ptr(self.knots, _dp), _ptr(cons, _dp), len(cons),
                                      _ptr(S, _dp), C.c_int64(S.shape[0]), C.c_int64(S.shape[1]), _ptr(out, _dp))
        return out


def first_min_argmin(e):
    e = np.ascontiguousarray(e)
    if e.dtype == np.float32:
        m = C.c_float()
        i = lib().orc_first_min_argmin_f32(_ptr(e, _fp), C.c_int64(len(e)), C.byref(m))
    else:
        e = _d(e)
        m = C.c_double()
        i = lib().orc_first_min_argmin_f64(_ptr(e, _dp), C.c_int64(len(e)), C.byref(m))
    return int(i), float(m.value)
