#!/bin/bash
# round 4, GPU session O: the wide shapes' apply launch with nothing beside it; contract tests with the new prologue
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r4o; mkdir -p $O
for sh in "4096 128" "1024 512"; do set -- $sh
  ALONE=1 BATCH=$1 WIDTH=$2 timeout 600 python tools/shape_bench.py 2>/dev/null | grep -v '^{' >> $O/alone.txt
done
timeout 900 python -m pytest tests/test_gpu_bench_contract.py -x -q -m gpu > $O/t_contract.log 2>&1; echo "contract rc $?" >> $O/rc.txt
timeout 600 python bench.py --steps 20 --warmup 5 > $O/bench_short.json 2> $O/bench_short.err
cat $O/alone.txt $O/rc.txt; tail -3 $O/t_contract.log
