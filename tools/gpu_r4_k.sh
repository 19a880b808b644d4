#!/bin/bash
# round 4, GPU session K: cache with the sort a batch early (parity + the tier's line), world-1 sharded step after the host trims
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r4k; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_cache.py -x -q -m gpu > $O/t_cache.log 2>&1; echo "cache rc $?" >> $O/rc.txt
timeout 900 python bench.py --no-cpu-baseline --no-laia --no-cold-tier --no-wide > $O/bench_cache.json 2> $O/bench_cache.err
export MASTER_ADDR=127.0.0.1 MASTER_PORT=29633 RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 HA_FORCE_SHARDED=1
timeout 900 python bench.py --steps 500 --warmup 60 --no-cpu-baseline 2>$O/sharded.err | grep '^{' > $O/bench_sharded_world1.json
timeout 600 python tools/framed_hostprof.py 2>&1 | grep "us/step" > $O/sharded_world1_host_vs_wall.txt
unset HA_FORCE_SHARDED RANK WORLD_SIZE LOCAL_RANK
timeout 900 python -m pytest tests/test_gpu_framed.py tests/test_gpu_bench_contract.py -x -q -m gpu > $O/t_framed.log 2>&1; echo "framed+contract rc $?" >> $O/rc.txt
cat $O/rc.txt $O/sharded_world1_host_vs_wall.txt; tail -3 $O/t_cache.log
python - <<'PY'
import json
d = json.loads([l for l in open("gpurun_out/r4k/bench_cache.json") if l.startswith("{")][-1])
print("cache tier", d.get("cache_tier", {}).get("us_per_step"), d["ms_per_step"])
d = json.loads(open("gpurun_out/r4k/bench_sharded_world1.json").readline())
print("sharded world1", d["ms_per_step"], d.get("config_c", {}).get("ms_per_step"))
PY
