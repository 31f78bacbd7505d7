#!/usr/bin/env python3
"""A/B of builds of libmg_hip.so (and, for the diagnostic build, of MG_DEBUG_FLAGS values) inside ONE process on the
SAME output buffer, variants interleaved round by round (cdna_hip_programming.md, methodology rule 24).

    python3 tools/ab.py VARIANT [VARIANT ...] [--rounds 6] [--steps 300] [--plain] [--two-launch]
    VARIANT = path/to/lib.so[:debug_flags[:kernel[:window[:arena_MiB[:root]]]]]   root: MG_OPT_ROOT_MODE (0, 1 float64, 2 split, 3 gate); kernel: 0 = by batch size, 1 = tile-major, 2 = chunk-stationary;
              window: MG_OPT_CHUNK_WINDOW (basis functions per time chunk, 0 = the planner's choice)
              e.g.  morphablegraphs_amd/csrc/libmg_hip.so  build/lib_x.so::1  dbg.so:1  dbg.so:512:2  lib.so::2:7

The output buffer comes from mg_device_malloc_placed of the first variant's library (--plain: mg_device_malloc).
Prints per variant the median / min step time over the rounds and the spread."""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from morphablegraphs_amd import _capi, synthetic   # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("variants", nargs="+")
ap.add_argument("--rounds", type=int, default=6)
ap.add_argument("--steps", type=int, default=300)
ap.add_argument("--batch", type=int, default=8192)
ap.add_argument("--plain", action="store_true")
ap.add_argument("--slow", action="store_true", help="a slow-class output buffer: plain allocations are held until one probes slow")
ap.add_argument("--two-launch", action="store_true")
args = ap.parse_args()

B, L, F, D = args.batch, 40, 156, 79
data = synthetic.make_walk_primitive(seed=0)
libs = {}
var = []
for v in args.variants:
    parts = v.split(":")
    path, flags, kern = os.path.abspath(parts[0]), (parts[1] if len(parts) > 1 else ""), int(parts[2]) if len(parts) > 2 and parts[2] else 0
    window = int(parts[3]) if len(parts) > 3 and parts[3] else 0
    arena = int(parts[4]) if len(parts) > 4 and parts[4] else 0   # MiB per arena block for the primitive's constants (0: one hipMalloc each)
    root = int(parts[5]) if len(parts) > 5 and parts[5] else 0
    path = (path, window, arena)
    if path not in libs:
        lib = _capi.load_library(path[0])
        ctx = _capi.Context(0, lib=lib)
        ctx.set_option(_capi.MG_OPT_CHUNK_WINDOW, window)   # read when the primitive's canonical grid is planned
        if arena:
            ctx.arena_begin(arena << 20)
        libs[path] = (lib, ctx, _capi.Primitive(ctx, data))
        if arena:
            ctx.arena_end()
        print(v, libs[path][2].step_plan(B), flush=True)
    var.append((v, path, flags, kern, root))
ctx0 = libs[var[0][1]][1]
S = ctx0.upload(np.random.default_rng(0).standard_normal((B, L)).astype(np.float32))
lp = ctx0.malloc(B * 4)
if args.slow:
    ctx0.set_option(_capi.MG_OPT_PLAIN_MALLOC, 1)
    held, out = [], None
    for _ in range(40):
        b = ctx0.malloc(B * F * D * 4)
        info = ctx0.probe_placement(b)
        if info["ratio"] > 1.15:
            out = b
            print("output buffer: slow class", info, "after", len(held), "fast ones", flush=True)
            break
        held.append(b)
    for b in held:
        b.free()
    ctx0.set_option(_capi.MG_OPT_PLAIN_MALLOC, 0)
    if out is None:
        raise SystemExit("no slow-class buffer among 40 plain allocations")
else:
    out = ctx0.malloc(B * F * D * 4) if args.plain else ctx0.malloc_placed(B * F * D * 4)
    print("output buffer:", "plain" if args.plain else out.placement, flush=True)


def run(prim, n):
    for _ in range(n):
        if args.two_launch:
            prim.back_project_frames_dev(S, np.float32, B, L, out, path=_capi.MG_PATH_MFMA)
            prim.gmm_log_prob_dev(S, np.float32, B, L, lp, np.float32)
        else:
            prim.step_frames_and_logp_dev(S, np.float32, B, L, out, lp)


times = {v[0]: [] for v in var}
for r in range(args.rounds + 1):
    for name, path, flags, kern, root in var:
        lib, ctx, prim = libs[path]
        ctx.set_option(_capi.MG_OPT_FRAMES_KERNEL, kern)
        ctx.set_option(_capi.MG_OPT_ROOT_MODE, root)
        if flags:
            os.environ["MG_DEBUG_FLAGS"] = flags
        else:
            os.environ.pop("MG_DEBUG_FLAGS", None)
        run(prim, 30)
        ctx.synchronize()
        t0 = time.perf_counter()
        run(prim, args.steps)
        ctx.synchronize()
        if r > 0:   # round 0 warms the clocks
            times[name].append(1e6 * (time.perf_counter() - t0) / args.steps)
ref = None
for name, path, flags, kern, root in var:   # every variant must write the same bytes as the first one (of its root mode)
    lib, ctx, prim = libs[path]
    ctx.set_option(_capi.MG_OPT_FRAMES_KERNEL, kern)
    ctx.set_option(_capi.MG_OPT_ROOT_MODE, root)
    if flags:
        continue                       # ablations change the results by design
    os.environ.pop("MG_DEBUG_FLAGS", None)
    ctx0.lib.mg_memset(ctx0.handle, out.ptr, 0xff, B * F * D * 4)
    ctx0.synchronize()
    run(prim, 1)
    ctx.synchronize()
    got = (ctx0.download(out, (B * F * D,), np.float32), ctx0.download(lp, (B,), np.float32))
    if ref is None:
        ref = got
    else:
        print("%-60s frames %s  log p %s" % (name, "identical" if np.array_equal(ref[0], got[0], equal_nan=True) else "DIFFER",
                                             "identical" if np.array_equal(ref[1], got[1], equal_nan=True) else "DIFFER"))
for name, _, _, _, _ in var:
    t = np.array(times[name])
    print("%-60s median %.2f  min %.2f  max %.2f us" % (name, np.median(t), t.min(), t.max()))
