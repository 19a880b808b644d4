#!/bin/bash
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r4a2; mkdir -p $O
timeout 1800 python -m pytest tests/test_gpu_laia.py tests/test_gpu_laia_config_d.py tests/test_gpu_example_wdl.py -x -q -m gpu > $O/t_laia.log 2>&1; echo "laia rc $?" >> $O/rc.txt
for a in 0 1 0 1; do HA_LAIA_AHEAD=$a timeout 600 python tools/laia_profile.py 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('ahead', d['one_batch_ahead'], 'in-call %.1f us' % d['us_per_global_batch'], 'thread wall %.1f us' % d['thread_wall_us_per_global_batch'])" >> $O/laia.txt; done
cat $O/rc.txt $O/laia.txt; tail -3 $O/t_laia.log
