"""The fused step N times on one placed buffer, for counter passes: python3 tools/probes/run_step.py <lib.so> [n] (MG_DEBUG_FLAGS from the environment
when the library is the diagnostic build)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from morphablegraphs_amd import _capi, synthetic  # noqa: E402

lib = _capi.load_library(os.path.abspath(sys.argv[1]))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 300
ctx = _capi.Context(0, lib=lib)
ctx.set_option(_capi.MG_OPT_FRAMES_KERNEL, 2)
prim = _capi.Primitive(ctx, synthetic.make_walk_primitive(seed=0))
B = 8192
S = ctx.upload(np.random.default_rng(0).standard_normal((B, 40)).astype(np.float32))
out, lp = ctx.malloc_placed(B * 156 * 79 * 4), ctx.malloc(B * 4)
for _ in range(n):
    prim.step_frames_and_logp_dev(S, np.float32, B, 40, out, lp)
ctx.synchronize()
