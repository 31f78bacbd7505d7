#!/usr/bin/env python3
"""Per-wave phase timers of the persistent frames kernel (diagnostic build only):
    make -C morphablegraphs_amd/csrc libmg_hip_dbg.so && python3 tools/stamp_run.py [flags ...]
Each argument is a MG_DEBUG_FLAGS value (16 is added); STAMP_B = batch, FUSED=0 for the stand-alone frames kernel,
PLAIN=1 for an unplaced output buffer, KERNEL=2 for the chunk-stationary kernel."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from morphablegraphs_amd import _capi, synthetic   # noqa: E402

lib = _capi.load_library(os.path.join(ROOT, "morphablegraphs_amd", "csrc", "libmg_hip_dbg.so"))
ctx = _capi.Context(0, lib=lib)
ctx.set_option(_capi.MG_OPT_FRAMES_KERNEL, int(os.environ.get("KERNEL", "0")))   # 2: the chunk-stationary kernel
prim = _capi.Primitive(ctx, synthetic.make_walk_primitive(seed=0))
B = int(os.environ.get("STAMP_B", "8192"))
S = ctx.upload(np.random.default_rng(0).standard_normal((B, 40)).astype(np.float32))
out = ctx.malloc(B * 156 * 79 * 4) if os.environ.get("PLAIN") else ctx.malloc_placed(B * 156 * 79 * 4)
print("output:", out.placement)
logp = ctx.malloc(B * 4)
for flags in (sys.argv[1:] or ["0"]):
    lite = bool(int(flags) & 32768)   # light stamps: the product kernel's timeline (no stamp stores inside the kernel's life)
    os.environ["MG_DEBUG_FLAGS"] = str(int(flags) if lite else int(flags) | 16)
    for _ in range(200):
        if os.environ.get("FUSED", "1") != "0":
            prim.step_frames_and_logp_dev(S, np.float32, B, 40, out, logp)
        else:
            prim.back_project_frames_dev(S, np.float32, B, 40, out, path=_capi.MG_PATH_MFMA)
    ctx.synchronize()
    ctx.profile_reset()
    ctx.profile_enable(True)
    for _ in range(50):
        if os.environ.get("FUSED", "1") != "0":
            prim.step_frames_and_logp_dev(S, np.float32, B, 40, out, logp)
        else:
            prim.back_project_frames_dev(S, np.float32, B, 40, out, path=_capi.MG_PATH_MFMA)
    ctx.synchronize()
    ms, n = ctx.profile_get("frames")
    ctx.profile_enable(False)
    print("==== MG_DEBUG_FLAGS = %s (+16): kernel %.2f us by its dispatch's own events (%d launches)" % (flags, 1e3 * ms / max(n, 1), n), flush=True)
    if lite:
        lib.mg_debug_dump_lite()
    else:
        lib.mg_debug_dump_stamps()
    sys.stdout.flush()
