"""Planner step (BASELINE configs[2]: 16 options x 4096 candidates) on builds of libmg_hip.so: wall time per step and the
fused kernel's own duration (dispatch events).   python3 tools/opt_step_ab.py [lib.so[@OPTIONS_STEP] ...]
(@2: MG_OPT_OPTIONS_STEP 2 = the counts kernel in front of every step, never the counts the step before drew ahead)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from morphablegraphs_amd import _capi, synthetic
from morphablegraphs_amd.motion_state_graph import HipPrimitiveSet
N = int(os.environ.get("N", "4096"))
DEVC = os.environ.get("DEVC", "1") == "1"     # component counts drawn on the device (bench.py --config graph does)
prims = synthetic.make_graph_primitives(int(os.environ.get("NOPT", "16")))
names = [p["name"] for p in prims]
cons = {nm: [{"type": "position", "t": float(p["n_canonical_frames"] - 1), "weight": 1.0, "target": [10.0, None, 5.0]},
             {"type": "direction", "t": float(p["n_canonical_frames"] - 1), "weight": 1.0, "target": [0.5, 1.0]}] for nm, p in zip(names, prims)}
for arg in (sys.argv[1:] or [None]):
    path, _, optstep = (arg or "").partition("@")
    ctx = _capi.Context(0, lib=_capi.load_library(os.path.abspath(path))) if path else _capi.Context(0)
    if optstep:
        ctx.set_option(_capi.MG_OPT_OPTIONS_STEP, int(optstep))
    pset = HipPrimitiveSet(prims, context=ctx)
    for i in range(100): pset.evaluate_options_on_device(names, cons, N, seed=i, device_counts=DEVC)
    t0 = time.perf_counter()
    for i in range(1000): pset.evaluate_options_on_device(names, cons, N, seed=i, device_counts=DEVC)
    wall = (time.perf_counter() - t0) / 1000 * 1e6
    ctx.profile_reset(); ctx.profile_enable(1)
    for i in range(200): pset.evaluate_options_on_device(names, cons, N, seed=i, device_counts=DEVC)
    ctx.profile_enable(0)
    ms, cnt = ctx.profile_get("options_step")
    print("%-28s step %.1f us, fused kernel %.1f us (%d launches)" % (arg or "default", wall, 1e3 * ms / max(cnt, 1), cnt), flush=True)
