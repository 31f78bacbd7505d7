"""Event-timed averages of the secondary kernels at the bench's batch size (8192 'walk' candidates):
log-likelihood, its Jacobian, fused keyframe scoring (plain / residual matrix / with FK chains), argmin,
device sampler.  Usage: python3 tools/micro_bench.py [batch]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from morphablegraphs_amd import _capi, synthetic

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
ctx = _capi.Context(0)
data = synthetic.make_walk_primitive(seed=0)
prim = _capi.Primitive(ctx, data)
S = np.random.default_rng(0).standard_normal((B, 40)).astype(np.float32)
d_S = ctx.upload(S)
d_lp = ctx.malloc(B * 8)
d_err = ctx.malloc(B * 8)
d_res = ctx.malloc(B * 8 * 8)
d_jac = ctx.malloc(B * 40 * 8)
d_x = ctx.malloc(B * 40 * 4)
d_comp = ctx.malloc(B * 4)
joints, animated = synthetic.make_skeleton()
sk = _capi.Skeleton(joints, animated)
cons = [{"type": "position", "t": 155.0, "weight": 1.0, "target": [40.0, None, -30.0]},
        {"type": "direction", "t": 155.0, "weight": 1.0, "target": [0.5, 1.0]}]
fk = [{"type": "joint_position", "joint": "LeftHand", "t": 155.0, "weight": 1.0, "target": [30.0, 90.0, -20.0]},
      {"type": "joint_position", "joint": "RightFoot", "t": 77.0, "weight": 1.0, "target": [None, 0.0, None]}]
cs = _capi.ConstraintSet(prim, cons)
cs_fk = _capi.ConstraintSet(prim, cons + fk, sk)
lib = prim.lib
C = _capi.C
counts = np.full(8, B // 8, dtype=np.int64)


def timed(name, fn, n=200):
    for _ in range(20):
        fn()
    ctx.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    ctx.synchronize()
    print("%-44s %8.2f us / launch (host wall clock over %d back-to-back launches)" % (name, 1e6 * (time.perf_counter() - t0) / n, n))


timed("gmm_log_prob (f32 in, f32 out)", lambda: prim.gmm_log_prob_dev(d_S, np.float32, B, 40, d_lp, np.float32))
timed("gmm_log_prob_jac", lambda: _capi._check(lib.mg_gmm_log_prob_jac(prim.handle, d_S.ptr, _capi.MG_F32, B, 40, d_jac.ptr)))
timed("score_constraints (2 root constraints)", lambda: prim.score_constraints_dev(cs, d_S, np.float32, B, 40, d_err, np.float64))
timed("score_constraint_residuals (2 root + 2 FK)", lambda: _capi._check(lib.mg_score_constraint_residuals(prim.handle, cs_fk.handle, d_S.ptr, _capi.MG_F32, B, 40, d_res.ptr)))
timed("argmin_first_dev", lambda: _capi._check(lib.mg_argmin_first_dev(ctx.handle, d_err.ptr, _capi.MG_F64, B, d_lp.ptr)))
timed("gmm_sample (device Philox)", lambda: _capi._check(lib.mg_gmm_sample(prim.handle, B, counts.ctypes.data_as(C.c_void_p), C.c_uint64(1), d_x.ptr, _capi.MG_F32, 40, d_comp.ptr)))
