#!/bin/bash
# per-kernel averages (rocprofv3 --kernel-trace --stats) of the default bench command for variant libraries ab/v/libherald_amd_<name>.so
O=$GRAFT_REPO_ROOT/gpurun_out/abk; mkdir -p $O
ARGS="--steps 512 --warmup 64 --no-cpu-baseline --no-kernel-pass --no-cache-tier --no-laia --no-cold-tier --no-wide"
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT; cp herald_amd/libherald_amd.so /tmp/lib_head.so
for t in "$@"; do
  if [ "$t" = head ]; then cp /tmp/lib_head.so herald_amd/libherald_amd.so; else cp ab/v/libherald_amd_$t.so herald_amd/libherald_amd.so; fi
  cd /tmp; rm -rf $O/$t
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/$t -- python3 $GRAFT_REPO_ROOT/bench.py $ARGS > $O/$t.log 2>&1
  f=$(find $O/$t -name "*kernel_stats.csv" | head -1)
  echo "== $t  $(grep '^{' $O/$t.log | python3 -c 'import json,sys; print("us/step %.2f" % (json.loads(sys.stdin.read())["ms_per_step"]*1e3))')"; python3 - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if "ha::q" in r["Name"] and "order_check" not in r["Name"]:
        print("   %-50s calls %5s avg %9.1f ns  min %8s" % (r["Name"][:50], r["Calls"], float(r["AverageNs"]), r["MinNs"]))
PY
  cd $GRAFT_REPO_ROOT
done
cp /tmp/lib_head.so herald_amd/libherald_amd.so
find $O -name "*.csv" -size +3M -delete
