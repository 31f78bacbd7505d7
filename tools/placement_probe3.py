"""Which allocation sizes land in the fast placement on this box?  Separate allocations, the 404 MB output at offset 0."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from morphablegraphs_amd import _capi, synthetic
ctx = _capi.Context(0)
prim = _capi.Primitive(ctx, synthetic.make_walk_primitive(seed=0))
B, L = 8192, 40
NB = B * 156 * 79 * 4
S = ctx.upload(np.random.default_rng(0).standard_normal((B, L)).astype(np.float32))
lp = ctx.malloc(B * 4)
def run(ptr, n=400):
    for _ in range(40): prim.step_frames_and_logp_dev(S, np.float32, B, L, ptr, lp)
    ctx.synchronize(); t0 = time.perf_counter()
    for _ in range(n): prim.step_frames_and_logp_dev(S, np.float32, B, L, ptr, lp)
    ctx.synchronize()
    return 1e6 * (time.perf_counter() - t0) / n
warm = ctx.malloc(NB); run(warm.ptr.value, 800); warm.free()
for extra in (0, 4096, 65536, 1 << 20, (1 << 21) - (NB % (1 << 21)), 3 << 20, 16 << 20, 64 << 20):
    bufs = [ctx.malloc(NB + extra) for _ in range(6)]
    print("size = output + %8d B (%.2f MiB total):" % (extra, (NB + extra) / 2**20), " ".join("%.1f" % run(b.ptr.value) for b in bufs), flush=True)
    for b in bufs: b.free()
# keep allocations alive while allocating more: does exhausting a region change the mode?
held = []
line = []
for k in range(16):
    b = ctx.malloc(NB); held.append(b); line.append("%.1f" % run(b.ptr.value, 250))
print("16 allocations held at once:", " ".join(line))
