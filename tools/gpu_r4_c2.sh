#!/bin/bash
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r4c2; mkdir -p $O
timeout 2400 python -m pytest tests/test_gpu_cache.py tests/test_gpu_cache_remote.py tests/test_plugins.py -x -q -m gpu > $O/t_cache.log 2>&1; echo "cache rc $?" >> $O/rc.txt
timeout 1500 python -m pytest tests/test_gpu_fullscale.py -x -q -m gpu -k "cache or cold" > $O/t_full.log 2>&1; echo "fullscale cache rc $?" >> $O/rc.txt
for i in 1 2 3; do
timeout 900 python bench.py --no-cpu-baseline --no-laia --no-wide 2>/dev/null | grep '^{' | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('cache tier %.2f us  cold tier %.2f us' % (d['cache_tier']['us_per_step'], d['cold_tier']['us_per_step']))" >> $O/tier.txt
done
cat $O/rc.txt $O/tier.txt; tail -3 $O/t_cache.log
