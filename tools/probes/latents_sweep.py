"""Probe: the frames kernels off the bench's shape -- primitives with 40 .. 64 latents, float32 and float64
latents, chunk-stationary (MG_OPT_FRAMES_KERNEL 2) and tile-major (1): us per launch of mg_back_project_frames (8192 candidates, 156
frames, 79 channels; algorithmic bytes B (4 L + 4 F D) + constants) and the fraction of 8 TB/s.  usage: python tools/probes/latents_sweep.py"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from morphablegraphs_amd import _capi, synthetic  # noqa: E402

ctx = _capi.Context(0)
B, F, D = 8192, 156, 79
out = ctx.malloc_placed(B * F * D * 4)
print("output:", ctx.placement_info(out))
for L in (40, 48, 52, 56, 64):
    prim = _capi.Primitive(ctx, synthetic.make_primitive(seed=0, n_components=L, n_frames=F, n_dim=D, n_gmm=8, name="walk%d" % L))
    for dtype in (np.float32, np.float64):
        S = ctx.upload(np.random.default_rng(0).standard_normal((B, L)).astype(dtype))
        row = []
        ref = None
        for kern in (2, 1):
            ctx.set_option(_capi.MG_OPT_FRAMES_KERNEL, kern)
            try:
                for _ in range(30):
                    prim.back_project_frames_dev(S, dtype, B, L, out, path=_capi.MG_PATH_MFMA)
                ctx.synchronize()
                t0 = time.perf_counter()
                for _ in range(200):
                    prim.back_project_frames_dev(S, dtype, B, L, out, path=_capi.MG_PATH_MFMA)
                ctx.synchronize()
                us = 1e6 * (time.perf_counter() - t0) / 200
                got = ctx.download(out.ptr.value + 777 * F * D * 4, (F * D,), np.float32)
                same = "" if ref is None else ("same bits" if np.array_equal(got.view(np.uint32), ref.view(np.uint32)) else "DIFFER")
                ref = got
                alg = B * (np.dtype(dtype).itemsize * L + 4 * F * D) + 4 * (31 * D * L + 31 * D + 4 * F)
                row.append("%s %7.2f us = %.3f %s" % ("chunk-stationary" if kern == 2 else "tile-major", us, alg / (us * 1e-6) / 8e12, same))
            except _capi.MGError as e:
                row.append("%s: %s" % ("chunk-stationary" if kern == 2 else "tile-major", str(e)[:60]))
        ctx.set_option(_capi.MG_OPT_FRAMES_KERNEL, 0)
        plan = prim.step_plan(B, out)
        print("L = %2d KK = %2d %s latents: %s | %s | default: %s" % (L, prim.kk, np.dtype(dtype).name, row[0], row[1], plan["kernel"]), flush=True)
        S.free()
    prim.close()
