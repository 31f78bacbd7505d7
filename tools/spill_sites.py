#!/usr/bin/env python3
"""Where a kernel's register spills sit: scratch loads / stores inside loops (backward branches) or outside (one-time values parked
around a role switch).   python3 tools/spill_sites.py <object or .so> <kernel-name regex>"""
import re
import subprocess
import sys
import tempfile
import os

obj, pat = sys.argv[1], re.compile(sys.argv[2])
llvm = "/opt/rocm/lib/llvm/bin"
with tempfile.TemporaryDirectory() as tmp:
    subprocess.check_call([llvm + "/llvm-objcopy", "--dump-section", ".hip_fatbin=" + tmp + "/fat.bin", obj])
    subprocess.check_call([llvm + "/clang-offload-bundler", "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--input=" + tmp + "/fat.bin",
                           "--output=" + tmp + "/dev.co", "--unbundle"])
    lines = subprocess.check_output([llvm + "/llvm-objdump", "-d", tmp + "/dev.co"]).decode().split("\n")
starts = [i for i, l in enumerate(lines) if re.match(r"^[0-9a-f]+ <", l)]
for si, st in enumerate(starts):
    name = lines[st].split("<")[1].rstrip(">:")
    if not pat.search(name):
        continue
    body = lines[st + 1:(starts[si + 1] if si + 1 < len(starts) else len(lines))]
    ins = []
    for l in body:
        m = re.match(r"\s+(\S+)\s+(.*?)//\s*([0-9A-F]+):", l)
        if m:
            ins.append((int(m.group(3), 16), m.group(1), m.group(2)))
    idx = {a: i for i, (a, _, _) in enumerate(ins)}
    loops = []
    for i, (a, op, args) in enumerate(ins):
        if op.startswith("s_cbranch") or op == "s_branch":
            m = re.search(r"(\d+)", args)
            if m:
                simm = int(m.group(1))
                simm = simm - 65536 if simm >= 32768 else simm
                t = a + 4 + simm * 4
                if t < a and t in idx:
                    loops.append((idx[t], i))
    sl = [i for i, (_, op, _) in enumerate(ins) if op.startswith("scratch_load")]
    ss = [i for i, (_, op, _) in enumerate(ins) if op.startswith("scratch_store")]
    inl = lambda i: any(a <= i <= b for a, b in loops)
    print("%s\n   %d instructions, %d loops; scratch loads %d (%d inside loops), scratch stores %d (%d inside loops)" % (
        name[:100], len(ins), len(loops), len(sl), sum(map(inl, sl)), len(ss), sum(map(inl, ss))))
