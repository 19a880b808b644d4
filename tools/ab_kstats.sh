#!/bin/bash
# same-box per-kernel averages (rocprofv3 --kernel-trace --stats) of the default bench command for trees ab/<name> and HEAD
O=$GRAFT_REPO_ROOT/gpurun_out/abk; mkdir -p $O
ARGS="--steps 512 --warmup 64 --no-cpu-baseline --no-kernel-pass --no-cache-tier --no-laia --no-cold-tier --no-wide"
export TMPDIR=/tmp
for t in "$@"; do
  if [ "$t" = head ]; then d=$GRAFT_REPO_ROOT; else d=$GRAFT_REPO_ROOT/ab/$t; fi
  cd /tmp; rm -rf $O/$t
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/$t -- python3 $d/bench.py $ARGS > $O/$t.log 2>&1
  f=$(find $O/$t -name "*kernel_stats.csv" | head -1)
  echo "== $t"; python3 - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if "ha::q" in r["Name"]:
        print("   %-50s calls %5s avg %9.1f ns  min %8s max %8s" % (r["Name"][:50], r["Calls"], float(r["AverageNs"]), r["MinNs"], r["MaxNs"]))
PY
done
find $O -name "*.csv" -size +3M -delete
