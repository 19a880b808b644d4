#!/bin/bash
# round 4, GPU session P: wide path with four small buckets per preparation workgroup: parity, shapes, per-kernel durations
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r4p; mkdir -p $O
timeout 2400 python -m pytest tests/test_gpu_qstep.py -x -q -m gpu -k "wide" > $O/t_wide.log 2>&1; echo "wide rc $?" >> $O/rc.txt
for sh in "4096 128" "1024 512" "2048 128"; do set -- $sh
  ALONE=1 BATCH=$1 WIDTH=$2 timeout 600 python tools/shape_bench.py 2>/dev/null | grep -v '^{' >> $O/shapes.txt
done
export TMPDIR=/tmp BATCH=4096 WIDTH=128
timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof -o s -- python3 tools/shape_bench.py > $O/prof.log 2>&1
unset BATCH WIDTH
timeout 2400 python -m pytest tests/test_gpu_qstep.py tests/test_gpu_fullscale.py -x -q -m gpu -k "not wide" > $O/t_rest.log 2>&1; echo "rest rc $?" >> $O/rc.txt
cat $O/rc.txt $O/shapes.txt; tail -3 $O/t_wide.log
