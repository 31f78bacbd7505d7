#!/bin/bash
# A/B of two MG_DEBUG_FLAGS values inside ONE process per round on the same output buffers (placement modes differ per
# buffer, so both variants are timed on each of 6 buffers): tools/flag_ab.sh <flagsA> <flagsB> [rounds]
for i in $(seq 1 ${3:-2}); do
  python3 - "$1" "$2" <<'PY'
import os, sys, time
import numpy as np
sys.path.insert(0, os.getcwd())
from morphablegraphs_amd import _capi, synthetic
fa, fb = sys.argv[1], sys.argv[2]
ctx = _capi.Context(0)
B, L = 8192, 40
S = ctx.upload(np.random.default_rng(0).standard_normal((B, L)).astype(np.float32))
lp = ctx.malloc(B * 4)
bufs = [ctx.malloc(B * 156 * 79 * 4) for _ in range(10)]
res = {}
for flags in (fa, fb, fa, fb):
    os.environ["MG_DEBUG_FLAGS"] = flags
    prim = _capi.Primitive(ctx, synthetic.make_walk_primitive(seed=0))   # the flags are read when the grid is built / launched
    row = []
    for b in bufs:
        for _ in range(100): prim.step_frames_and_logp_dev(S, np.float32, B, L, b, lp)
        ctx.synchronize(); t0 = time.perf_counter()
        for _ in range(600): prim.step_frames_and_logp_dev(S, np.float32, B, L, b, lp)
        ctx.synchronize(); row.append(1e6 * (time.perf_counter() - t0) / 600)
    res.setdefault(flags, []).append(row)
    prim.close()
for flags in (fa, fb):
    best = np.min(np.array(res[flags]), axis=0)
    print("flags %7s:" % flags, " ".join("%.1f" % v for v in best), " mean %.2f" % best.mean())
PY
done
