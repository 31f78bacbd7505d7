"""Step time of the REAL frames kernel (fused step, B = 8192) on output buffers assembled from physical chunks of several sizes
(mg_device_malloc_chunked) against plain hipMalloc buffers: does scattering the physical placement help the kernel as it helps
the pattern replica (tools/probes/placement_pmc.hip T2)?"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from morphablegraphs_amd import _capi, synthetic
ctx = _capi.Context(0)
prim = _capi.Primitive(ctx, synthetic.make_walk_primitive(seed=0))
B, L = 8192, 40
NB = B * 156 * 79 * 4
S = ctx.upload(np.random.default_rng(0).standard_normal((B, L)).astype(np.float32))
lp = ctx.malloc(B * 4)
def run(buf, n=400):
    for _ in range(60): prim.step_frames_and_logp_dev(S, np.float32, B, L, buf, lp)
    ctx.synchronize(); t0 = time.perf_counter()
    for _ in range(n): prim.step_frames_and_logp_dev(S, np.float32, B, L, buf, lp)
    ctx.synchronize()
    return 1e6 * (time.perf_counter() - t0) / n
w = ctx.malloc(NB); run(w, 1500); w.free()
for chunk in (0, 8 << 20, 16 << 20, 32 << 20, 64 << 20, 128 << 20, 0, 32 << 20):
    bufs = [ctx.malloc(NB, chunk_bytes=chunk) for _ in range(6)]
    print("%-22s" % ("plain hipMalloc" if chunk == 0 else "chunks of %d MiB" % (chunk >> 20)),
          " ".join("%.1f(%.2f)" % (run(b), ctx.probe_placement(b)["ratio"]) for b in bufs), flush=True)
    for b in bufs: b.free()
