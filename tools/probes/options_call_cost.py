"""mg_options_step: what the C call costs around its one kernel (launch, read-back, synchronisation)."""
import os, sys, time, ctypes as C
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from morphablegraphs_amd import synthetic, _capi
from morphablegraphs_amd.motion_state_graph import HipPrimitiveSet
prims = synthetic.make_graph_primitives(16)
names = [p["name"] for p in prims]
cons = {nm: [{"type": "position", "t": float(p["n_canonical_frames"] - 1), "weight": 1.0, "target": [10.0, None, 5.0]},
             {"type": "direction", "t": float(p["n_canonical_frames"] - 1), "weight": 1.0, "target": [0.5, 1.0]}] for nm, p in zip(names, prims)}
pset = HipPrimitiveSet(prims)
n = 4096
for i in range(50):
    pset.evaluate_options_on_device(names, cons, n, seed=i)
plan = pset._step_plan(tuple(names), n, np.dtype(np.float32))
steps = plan["steps"]
lib = steps[0][2].lib
m, stride, host = len(steps), plan["stride"], plan["host"]
def call():
    _capi._check(lib.mg_options_step(m, plan["prims"], plan["csets"], n, plan["cnts"], plan["seeds"], plan["xs"], _capi.MG_F32, plan["lds"],
                                     plan["errs"], plan["shared"].ptr, stride, host.ctypes.data_as(C.c_void_p)))
for _ in range(100): call()
t0 = time.perf_counter()
for _ in range(1000): call()
dt = (time.perf_counter() - t0) / 1000
ctx = pset.ctx
ctx.profile_reset(); ctx.profile_enable(1)
for _ in range(200): call()
ctx.profile_enable(0)
ms, cnt = ctx.profile_get("options_step")
t0 = time.perf_counter()
for i in range(1000): pset.evaluate_options_on_device(names, cons, n, seed=i)
dt2 = (time.perf_counter() - t0) / 1000
print("C call %.1f us, its kernel %.1f us, whole Python step %.1f us" % (1e6 * dt, 1e3 * ms / cnt, 1e6 * dt2))
