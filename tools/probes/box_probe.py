#!/usr/bin/env python3
"""What kind of box is this?  Fill and store-pattern times of placed and plain output buffers, clocks and power as rocm-smi sees them.
    python3 tools/probes/box_probe.py [n_buffers]"""
import os
import subprocess
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from morphablegraphs_amd import _capi   # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 6
ctx = _capi.Context(0)
print(ctx.device_info())
nbytes = 8192 * 156 * 79 * 4
bufs = []
for i in range(n):
    b = ctx.malloc(nbytes)
    bufs.append(b)
    print("plain  %d: %s" % (i, ctx.probe_placement(b)), flush=True)
for i in range(2):
    b = ctx.malloc_placed(nbytes)
    bufs.append(b)
    print("placed %d: %s" % (i, b.placement), flush=True)
for args in (["--showclocks"], ["--showpower"], ["--showmemuse"], ["--showperflevel"]):
    try:
        out = subprocess.run(["/opt/rocm/bin/rocm-smi"] + args, capture_output=True, text=True, timeout=30).stdout
        print("\n".join(l for l in out.splitlines() if l.strip() and "====" not in l)[:1500])
    except Exception as e:   # noqa: BLE001
        print("rocm-smi", args, "failed:", e)
