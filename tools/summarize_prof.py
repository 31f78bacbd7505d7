#!/usr/bin/env python3
"""Summarise a tools/prof_bench.sh output directory into one small JSON/markdown file
(per-kernel average duration from the kernel trace; PMC counters averaged per launch)."""
import csv, glob, json, os, sys
from collections import defaultdict

def main(d, out):
    res = {"dir": os.path.basename(d), "kernels": {}, "pmc": {}}
    for f in glob.glob(os.path.join(d, "trace", "**", "*kernel_trace.csv"), recursive=True):
        agg = defaultdict(list)
        for r in csv.DictReader(open(f)):
            agg[r["Kernel_Name"]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
        for k, v in agg.items():
            v2 = v[len(v) // 10:]  # drop warm-up launches
            res["kernels"][k[:90]] = {"launches": len(v), "avg_us": sum(v2) / len(v2), "min_us": min(v), "max_us": max(v)}
            if len(v) >= 12000:   # bench.py's default run since round 5: five timed windows of 2000 steps, then the events pass of 2000 (before them: probes, ramp, warm-up)
                res["kernels"][k[:90]]["timed_windows_avg_us"] = sum(v[-12000:-2000]) / 10000.0
                res["kernels"][k[:90]]["events_pass_avg_us"] = sum(v[-2000:]) / 2000.0
            elif len(v) >= 2000:
                res["kernels"][k[:90]]["timed_region_avg_us"] = sum(v[-2000:]) / 2000.0
    for f in glob.glob(os.path.join(d, "pmc_*", "**", "*counter_collection.csv"), recursive=True):
        agg = defaultdict(lambda: defaultdict(list))
        for r in csv.DictReader(open(f)):
            agg[r["Kernel_Name"][:90]][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, cs in agg.items():
            for c, v in cs.items():
                res["pmc"].setdefault(k, {})[c] = sum(v) / len(v)
    json.dump(res, open(out, "w"), indent=1, sort_keys=True)
    print(json.dumps(res, indent=1, sort_keys=True))

if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
