#!/usr/bin/env python3
"""Where a planner step's wall time goes (configs[2]: 16 options x 4096 candidates): the Python method, the C call alone
(host counts / device counts), the C call without read-back + a separate synchronisation, the kernel by its own events."""
import ctypes as C
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from morphablegraphs_amd import _capi, synthetic   # noqa: E402
from morphablegraphs_amd.motion_state_graph import HipPrimitiveSet   # noqa: E402

n = 4096
prims = synthetic.make_graph_primitives(16)
names = [p["name"] for p in prims]
cons = {nm: [{"type": "position", "t": float(p["n_canonical_frames"] - 1), "weight": 1.0, "target": [10.0, None, 5.0]},
             {"type": "direction", "t": float(p["n_canonical_frames"] - 1), "weight": 1.0, "target": [0.5, 1.0]}] for nm, p in zip(names, prims)}
pset = HipPrimitiveSet(prims)
for dc in (False, True):
    for i in range(50):
        pset.evaluate_options_on_device(names, cons, n, seed=i, device_counts=dc)
    t = time.perf_counter()
    for i in range(1000):
        pset.evaluate_options_on_device(names, cons, n, seed=i, device_counts=dc)
    print("python method, device_counts=%s: %.1f us per step" % (dc, (time.perf_counter() - t) / 1000 * 1e6))
plan = pset._step_plan(tuple(names), n, np.dtype(np.float32))
lib, m, stride = plan["lib"], len(names), plan["stride"]
code = _capi.MG_F32
ctx = pset.ctx


def c_host(readback=True):
    return lib.mg_options_step(m, plan["prims"], plan["csets"], n, plan["cnts"], plan["seeds"], plan["xs"], code, plan["lds"], plan["errs"],
                               plan["shared_ptr"], stride, plan["host_ptr"] if readback else None)


def c_dev(readback=True):
    return lib.mg_options_step_device_counts(m, plan["prims"], plan["csets"], n, plan["seeds"], plan["xs"], code, plan["lds"], plan["errs"],
                                             plan["shared_ptr"], stride, plan["host_ptr"] if readback else None, None)


for name, fn in (("mg_options_step (counts given)", c_host), ("mg_options_step_device_counts", c_dev)):
    for rb in (True, False):
        for i in range(50):
            fn(rb)
        ctx.synchronize()
        t = time.perf_counter()
        for i in range(1000):
            fn(rb)
        ctx.synchronize()
        print("%s, read-back %s: %.1f us per call" % (name, rb, (time.perf_counter() - t) / 1000 * 1e6))
t = time.perf_counter()
for i in range(1000):
    for k, st in enumerate(plan["steps"]):
        plan["counts"][k, :len(st[8])] = np.random.multinomial(n, st[8])
print("16 x np.random.multinomial: %.1f us" % ((time.perf_counter() - t) / 1000 * 1e6))
t = time.perf_counter()
for i in range(1000):
    ctx.synchronize()
print("synchronize of an idle stream: %.1f us" % ((time.perf_counter() - t) / 1000 * 1e6))
ctx.profile_reset()
ctx.profile_enable(1)
for i in range(200):
    c_dev(True)
ms, cnt = ctx.profile_get("options_step")
print("mg_options_fused_kernel by its own events: %.1f us (%d launches)" % (1e3 * ms / max(cnt, 1), cnt))
