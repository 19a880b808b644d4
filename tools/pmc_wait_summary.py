#!/usr/bin/env python3
"""Condense the counter passes of tools/pmc_wait_counters.sh: per kernel and counter, the per-dispatch mean over the
dispatches of the pass.  usage: pmc_wait_summary.py <gpurun_out/pmcw_TAG> <out.json>"""
import csv, glob, json, os, sys

src, dst = sys.argv[1], sys.argv[2]
csv.field_size_limit(1 << 30)
acc = {}
for f in sorted(glob.glob(os.path.join(src, "g*", "*", "*_counter_collection.csv"))):
    with open(f) as fh:
        for r in csv.DictReader(fh):
            name = r["Kernel_Name"]
            if "ha::" not in name:
                continue
            name = name.replace("void ", "").split("(")[0]
            s = acc.setdefault(name, {}).setdefault(r["Counter_Name"], [0.0, 0])
            s[0] += float(r["Counter_Value"])
            s[1] += 1
out = {"note": "rocprofv3 --pmc passes, one group per run (tools/pmc_wait_counters.sh); per-dispatch means; SQ_*_CYCLES in "
               "quad-cycles summed over the chip's SQs as rocprofv3 reports them",
       "groups": open(os.path.join(src, "groups.txt")).read().splitlines() if os.path.exists(os.path.join(src, "groups.txt")) else [],
       "kernels": {k: {c: {"mean": v[0] / v[1], "dispatches": v[1]} for c, v in sorted(d.items())} for k, d in acc.items()}}
with open(dst, "w") as fh:
    json.dump(out, fh, indent=1)
q = out["kernels"].get("ha::qapply_kernel", {})
for c, v in q.items():
    print("%-40s %16.1f  (%d dispatches)" % (c, v["mean"], v["dispatches"]))
