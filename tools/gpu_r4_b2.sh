#!/bin/bash
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r4b2; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_bench_contract.py tests/test_plugins.py -x -q -m gpu > $O/t.log 2>&1; echo "rc $?" >> $O/rc.txt
cat $O/rc.txt; tail -25 $O/t.log
