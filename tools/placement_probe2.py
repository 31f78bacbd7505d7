"""Step time against where the 404 MB output sits: windows inside ONE big allocation (offsets of 2 MiB multiples and odd
ones), and separate allocations of several sizes."""
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
def run(ptr, n=600):
    for _ in range(60): prim.step_frames_and_logp_dev(S, np.float32, B, L, ptr, lp)
    ctx.synchronize(); t0 = time.perf_counter()
    for _ in range(n): prim.step_frames_and_logp_dev(S, np.float32, B, L, ptr, lp)
    ctx.synchronize()
    return 1e6 * (time.perf_counter() - t0) / n
big = ctx.malloc(6 * (1 << 30))
run(big.ptr.value, 800)
base = big.ptr.value
print("one 6 GiB allocation at %x" % base)
for off in [0, 512 << 20, 1024 << 20, 1536 << 20, 2048 << 20, 2560 << 20, 3072 << 20, 4096 << 20, 5000 << 20,
            (512 << 20) + (1 << 20), (512 << 20) + 4096, (512 << 20) + 128, (512 << 20) + 64, (512 << 20) + 16, (1024 << 20) + (1 << 20)]:
    print("  offset %5d MiB + %7d B: %.1f us" % (off >> 20, off & ((1 << 20) - 1), run(base + off)), flush=True)
big.free()
for size in (NB, NB + (1 << 21), 512 << 20, 1 << 30):
    bufs = [ctx.malloc(size) for _ in range(5)]
    print("separate allocations of %d MiB:" % (size >> 20), " ".join("%.1f" % run(b.ptr.value) for b in bufs), flush=True)
    for b in bufs: b.free()
