#!/usr/bin/env python3
"""Condense the rocprofv3 output of tools/profile_round.sh into the small files kept under profiles/:
  bench_kernel_stats.csv  the ha:: rows of the --kernel-trace --stats summary
  pmc_traffic.json        per-kernel HBM bytes per launch from the FETCH_SIZE / WRITE_SIZE passes
usage: pmc_summary.py <gpurun_out/prof_TAG> <out_dir>"""
import csv, glob, json, os, sys

src, dst = sys.argv[1], sys.argv[2]
os.makedirs(dst, exist_ok=True)
csv.field_size_limit(1 << 30)

rows = []
for f in glob.glob(os.path.join(src, "stats", "*", "*_kernel_stats.csv")):
    with open(f) as fh:
        rd = csv.DictReader(fh)
        fields = rd.fieldnames
        rows += [r for r in rd if "ha::" in r["Name"]]
if rows:
    with open(os.path.join(dst, "bench_kernel_stats.csv"), "w", newline="") as fh:
        w = csv.DictWriter(fh, fieldnames=fields)
        w.writeheader()
        w.writerows(rows)


def short(name):
    name = name.replace("void ", "")
    return name.split("(")[0]


acc = {}
for counter, sub in (("FETCH_SIZE", "pmc_fetch"), ("WRITE_SIZE", "pmc_write")):
    for f in glob.glob(os.path.join(src, sub, "*", "*_counter_collection.csv")):
        with open(f) as fh:
            for r in csv.DictReader(fh):
                if "ha::" not in r["Kernel_Name"] or r["Counter_Name"] != counter:
                    continue
                k = acc.setdefault(short(r["Kernel_Name"]), {})
                s = k.setdefault(counter, [0.0, 0])
                s[0] += float(r["Counter_Value"])
                s[1] += 1
out = {"note": "rocprofv3 --pmc passes (separate runs) of the bench command of the calling script (tools/profile_round.sh, tools/profile_sharded.sh); "
               "FETCH_SIZE / WRITE_SIZE are per-dispatch means in KB as reported; hbm_bytes_per_launch applies "
               "the gfx950 correction of MI355X_MICROARCH.md (FETCH_SIZE reports half of a wide coalesced "
               "read): 2*FETCH + WRITE",
       "kernels": {}}
for name, k in acc.items():
    if "FETCH_SIZE" in k and "WRITE_SIZE" in k:
        fe = k["FETCH_SIZE"][0] / k["FETCH_SIZE"][1]
        wr = k["WRITE_SIZE"][0] / k["WRITE_SIZE"][1]
        out["kernels"][name] = {"FETCH_SIZE_KB_mean": fe, "WRITE_SIZE_KB_mean": wr,
                                "dispatches": k["FETCH_SIZE"][1],
                                "hbm_bytes_per_launch": (2.0 * fe + wr) * 1024.0}
with open(os.path.join(dst, "pmc_traffic.json"), "w") as fh:
    json.dump(out, fh, indent=1)
print(json.dumps(out["kernels"], indent=1))
