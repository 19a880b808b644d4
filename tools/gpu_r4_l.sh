#!/bin/bash
# round 4, GPU session L: two preparation streams on the wide path: parity, shapes (A/B against one stream), block sweep
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r4l; mkdir -p $O
timeout 2400 python -m pytest tests/test_gpu_qstep.py -x -q -m gpu -k "wide" > $O/t_wide.log 2>&1; echo "wide rc $?" >> $O/rc.txt
for sh in "4096 128" "1024 512"; do set -- $sh
  for blk in 4 8; do
    BLOCK=$blk BATCH=$1 WIDTH=$2 timeout 600 python tools/shape_bench.py 2>/dev/null | head -1 | sed "s/^/two streams, block $blk: /" >> $O/shapes.txt
  done
  HA_QWIDE_ONE_STREAM=1 BATCH=$1 WIDTH=$2 timeout 600 python tools/shape_bench.py 2>/dev/null | head -1 | sed "s/^/one stream, block 4: /" >> $O/shapes.txt
done
SYNC=events BATCH=4096 WIDTH=128 timeout 600 python tools/shape_bench.py 2>/dev/null | head -1 | sed "s/^/two streams, events: /" >> $O/shapes.txt
timeout 2400 python -m pytest tests/test_gpu_qstep.py tests/test_gpu_fullscale.py -x -q -m gpu -k "not wide" > $O/t_rest.log 2>&1; echo "rest rc $?" >> $O/rc.txt
cat $O/rc.txt $O/shapes.txt; tail -3 $O/t_wide.log
