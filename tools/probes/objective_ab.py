"""The optimiser's objective in one launch (131072 candidates, two root constraints) on builds of libmg_hip.so, and the mixture alone:
kernel time by dispatch events.  usage: python tools/probes/objective_ab.py lib.so[@GMM_KERNEL_OPTION] ...
(@3: MG_OPT_GMM_KERNEL 3 = the channel rows by MFMAs of their own, not folded into the mixture's padding columns)"""
import os
import sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from morphablegraphs_amd import _capi, synthetic  # noqa: E402
B, L = 131072, 40
data = synthetic.make_walk_primitive(seed=0)
X = np.random.default_rng(0).standard_normal((B, L)).astype(np.float32)
for arg in sys.argv[1:]:
    path, _, opt = arg.partition("@")
    ctx = _capi.Context(0, lib=_capi.load_library(os.path.abspath(path)))
    if opt:
        ctx.set_option(_capi.MG_OPT_GMM_KERNEL, int(opt))
    prim = _capi.Primitive(ctx, data)
    cset = _capi.ConstraintSet(prim, [{"type": "position", "t": 155.0, "weight": 1.0, "target": [40.0, None, -30.0]},
                                      {"type": "direction", "t": 155.0, "weight": 1.0, "target": [0.5, 1.0]}])
    S, obj, lp = ctx.upload(X), ctx.malloc(B * 8), ctx.malloc(B * 4)
    out = []
    for what in ("objective", "mixture"):
        fn = (lambda: prim.objective_dev(cset, S, np.float32, B, L, 1.0, 1.0, obj_dev=obj)) if what == "objective" else (lambda: prim.gmm_log_prob_dev(S, np.float32, B, L, lp, np.float32))
        for _ in range(50):
            fn()
        ctx.synchronize(); ctx.profile_reset(); ctx.profile_enable(1)
        for _ in range(300):
            fn()
        ctx.synchronize(); ctx.profile_enable(False)
        ms, n = ctx.profile_get("gmm_log_prob")
        out.append("%s %.1f us" % (what, 1e3 * ms / max(n, 1)))
    print("%-28s %s" % (os.path.basename(arg), "   ".join(out)), flush=True)
