#!/bin/bash
# round 4, GPU session Q: per-kernel durations of the laia scheduler's global batch
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r4q; mkdir -p $O
export TMPDIR=/tmp
timeout 600 python tools/laia_profile.py > $O/laia_plain.json 2>/dev/null
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o laia -- python3 tools/laia_profile.py > $O/prof.log 2>&1
f=$(find $O/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/laia_kernel_stats.csv
find $O/prof -name "*kernel_trace.csv" -size +20M -delete
cat $O/laia_plain.json; head -20 $O/laia_kernel_stats.csv | cut -c1-150
