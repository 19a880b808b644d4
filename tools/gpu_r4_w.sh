#!/bin/bash
# round 4, GPU session W: new edge-case tests of the wide path
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r4w; mkdir -p $O
timeout 2400 python -m pytest tests/test_gpu_qstep.py -x -q -m gpu -k "quarter or wide" > $O/t.log 2>&1; echo "rc $?" >> $O/rc.txt
cat $O/rc.txt; tail -15 $O/t.log
