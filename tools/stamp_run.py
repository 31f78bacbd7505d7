import ctypes, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from morphablegraphs_amd import _capi, synthetic
ctx = _capi.Context(0)
prim = _capi.Primitive(ctx, synthetic.make_walk_primitive(seed=0))
B = int(os.environ.get("STAMP_B", "8192"))
S = ctx.upload(np.random.default_rng(0).standard_normal((B, 40)).astype(np.float32))
out = ctx.malloc(B * 156 * 79 * 4)
logp = ctx.malloc(B * 4)
for _ in range(5):
    if os.environ.get("FUSED"):
        prim.step_frames_and_logp_dev(S, np.float32, B, 40, out, logp)
    else:
        prim.back_project_frames_dev(S, np.float32, B, 40, out, path=_capi.MG_PATH_MFMA)
ctx.synchronize()
_capi.load_library().mg_debug_dump_stamps()
