"""Does PHYSICALLY CONTIGUOUS memory (hipExtMallocWithFlags(..., hipDeviceMallocContiguous)) belong to the fast class?  Plain hipMalloc
candidates of the bench's output size and contiguous ones, interleaved and held together, each probed twice with the library's own
placement probe (pattern us, pattern / fill ratio)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from morphablegraphs_amd import _capi
hip = C.CDLL("libamdhip64.so")
hip.hipExtMallocWithFlags.argtypes = [C.POINTER(C.c_void_p), C.c_size_t, C.c_uint]
hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
hip.hipFree.argtypes = [C.c_void_p]
ctx = _capi.Context(0)
nbytes = int(sys.argv[1]) if len(sys.argv) > 1 else 8192 * 156 * 79 * 4
n = int(sys.argv[2]) if len(sys.argv) > 2 else 12
FLAGS = [int(x, 0) for x in sys.argv[3].split(",")] if len(sys.argv) > 3 else [0x4]   # hipDeviceMallocContiguous 0x4, Finegrained 0x1, Uncached 0x3
NAMES = {0x4: "contiguous", 0x1: "finegrained", 0x3: "uncached"}
held = []
for i in range(2 * n):
    p = C.c_void_p()
    flag = FLAGS[(i // 2) % len(FLAGS)]
    kind = NAMES.get(flag, hex(flag)) if i % 2 else "plain"
    rc = hip.hipExtMallocWithFlags(C.byref(p), nbytes, flag) if i % 2 else hip.hipMalloc(C.byref(p), nbytes)
    if rc != 0:
        print("%2d %-10s allocation failed: hip status %d" % (i, kind, rc), flush=True)
        continue
    held.append(p)
    a, c = ctx.probe_placement(p.value, nbytes), ctx.probe_placement(p.value, nbytes)
    best = min(a["pattern_us"], c["pattern_us"])
    print("%2d %-10s %#x  pattern %.1f / %.1f us   ratio %.3f / %.3f   %.2f TB/s  %s" % (i, kind, p.value, a["pattern_us"], c["pattern_us"], a["ratio"], c["ratio"],
          nbytes / best * 1e-6, "FAST" if a["fast"] or c["fast"] else "slow"), flush=True)
for p in held:
    hip.hipFree(p)
