#!/bin/bash
# round 4, GPU session R: laia bits kernel with grouped round trips: parity + timing
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r4r; mkdir -p $O
export TMPDIR=/tmp
timeout 1800 python -m pytest tests/test_gpu_laia.py tests/test_gpu_laia_config_d.py -x -q -m gpu > $O/t_laia.log 2>&1; echo "laia rc $?" >> $O/rc.txt
for i in 1 2 3; do timeout 600 python tools/laia_profile.py 2>/dev/null >> $O/laia_plain.json; done
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o laia -- python3 tools/laia_profile.py > $O/prof.log 2>&1
f=$(find $O/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/laia_kernel_stats.csv
find $O/prof -name "*kernel_trace.csv" -size +20M -delete
cat $O/rc.txt; tail -2 $O/t_laia.log; cut -c1-60 $O/laia_plain.json; head -12 $O/laia_kernel_stats.csv | cut -c1-120
