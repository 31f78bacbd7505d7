#!/bin/bash
# two builds of libmg_hip.so on the same output buffers, one process per library but the SAME candidate order is no
# guarantee of the same placement -- so each library is timed on 8 buffers and the per-library minimum / median compared.
# usage: tools/lib_ab.sh ab/lib_a.so ab/lib_b.so [rounds]
for i in $(seq 1 ${3:-2}); do for L in $1 $2; do
MG_HIP_LIB=$PWD/$L python3 - "$L" <<'PY'
import os, sys, time
import numpy as np
sys.path.insert(0, os.getcwd())
from morphablegraphs_amd import _capi, synthetic
ctx = _capi.Context(0)
prim = _capi.Primitive(ctx, synthetic.make_walk_primitive(seed=0))
B, L = 8192, 40
S = ctx.upload(np.random.default_rng(0).standard_normal((B, L)).astype(np.float32))
lp = ctx.malloc(B * 4)
bufs = [ctx.malloc(B * 156 * 79 * 4) for _ in range(8)]
row = []
for b in bufs:
    for _ in range(150): prim.step_frames_and_logp_dev(S, np.float32, B, L, b, lp)
    ctx.synchronize(); t0 = time.perf_counter()
    for _ in range(800): prim.step_frames_and_logp_dev(S, np.float32, B, L, b, lp)
    ctx.synchronize(); row.append(1e6 * (time.perf_counter() - t0) / 800)
print("%-22s" % sys.argv[1], " ".join("%.1f" % v for v in row), " min %.1f median %.1f" % (min(row), float(np.median(row))))
PY
done; done
