#!/bin/bash
# rocprofv3 kernel breakdown of the cache tier (development aid)
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/cb; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/cb -- python3 $GRAFT_REPO_ROOT/tools/cache_bench.py > /tmp/cb.log 2>&1
grep "cache tier" /tmp/cb.log
python3 - <<PY
import csv,glob
for f in glob.glob("/tmp/cb/*/*_kernel_stats.csv"):
    rows=[r for r in csv.DictReader(open(f)) if "ha::" in r["Name"]]
    tot=sum(float(r["TotalDurationNs"]) for r in rows)
    for r in sorted(rows,key=lambda r:-float(r["TotalDurationNs"])):
        print("%-60s calls %5s avg %7.2f us  share %4.1f%%"%(r["Name"].replace("void ","")[:60],r["Calls"],float(r["AverageNs"])/1e3,100*float(r["TotalDurationNs"])/tot))
    print("total GPU us per step (464 steps):", tot/464/1e3)
PY
