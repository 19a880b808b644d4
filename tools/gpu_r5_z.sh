#!/bin/bash
O=$GRAFT_REPO_ROOT/gpurun_out/r5z; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
BATCH=4096 WIDTH=128 rocprofv3 --kernel-trace --output-format csv -d $O/tr -- python3 $GRAFT_REPO_ROOT/tools/shape_bench.py > $O/shape.log 2>&1
cd $GRAFT_REPO_ROOT
python3 tools/stream_timeline.py $O/tr 60 9 | tee $O/wide_timeline.txt
find $O -name "*.csv" -size +3M -delete
