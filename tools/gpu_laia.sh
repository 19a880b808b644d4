#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
R=$(pwd); O=$R/gpurun_out/laia; mkdir -p $O
python tools/laia_prof.py 2>&1 | grep batches
HA_LAIA_HOST=1 python tools/laia_prof.py 2>&1 | grep batches
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $R/tools/laia_prof.py > $O/stats.log 2>&1
python3 - <<PY
import csv, glob
for f in glob.glob("$O/stats/*/*_kernel_stats.csv"):
    for r in csv.DictReader(open(f)):
        print("%-80s calls %6s avg_us %9.2f total_ms %8.2f" % (r["Name"][:80], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6))
PY
