#!/bin/bash
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r4tol; mkdir -p $O
timeout 2400 python -m pytest tests/test_gpu_tolerance.py tests/test_gpu_framed.py tests/test_gpu_scatter.py -x -q -m gpu > $O/t.log 2>&1; echo "rc $?" >> $O/rc.txt
export MASTER_ADDR=127.0.0.1 MASTER_PORT=29633 RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 HA_FORCE_SHARDED=1
for i in 1 2; do
timeout 900 python bench.py --steps 500 --warmup 80 --no-cpu-baseline 2>/dev/null | grep '^{' | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('config A %.2f us  config_c %.2f us' % (d['ms_per_step']*1e3, d['config_c']['ms_per_step']*1e3)); print({k:round(v['us'],1) for k,v in (d['config_c']['roofline'].get('kernels') or {}).items()})" >> $O/sharded.txt
done
cat $O/rc.txt $O/sharded.txt; tail -3 $O/t.log
