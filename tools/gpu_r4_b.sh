#!/bin/bash
# round 4, GPU session B: yardstick variants, sized FramedStep + laia example tests, world-1 sharded bench
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4b; mkdir -p $O
timeout 400 tools/_bin/floor_bench 16 > $O/floor.txt 2>&1
timeout 1500 python -m pytest tests/test_gpu_framed.py -x -q -m gpu > $O/t_framed.log 2>&1; echo "framed rc $?" >> $O/rc.txt
timeout 1500 python -m pytest tests/test_gpu_example_wdl.py -x -q -m gpu > $O/t_example.log 2>&1; echo "example rc $?" >> $O/rc.txt
timeout 900 python -m pytest tests/test_gpu_sharded_multirank.py tests/test_gpu_hetu_ops.py tests/test_gpu_cache.py -x -q -m gpu > $O/t_misc.log 2>&1; echo "misc rc $?" >> $O/rc.txt
MASTER_ADDR=127.0.0.1 MASTER_PORT=29633 RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 HA_FORCE_SHARDED=1 timeout 900 python bench.py --steps 500 --warmup 50 --no-cpu-baseline > $O/bench_sharded_world1.json 2> $O/bench_sharded_world1.err
MASTER_ADDR=127.0.0.1 MASTER_PORT=29634 RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 HA_FORCE_SHARDED=1 HA_SHARD_FIXED=1 timeout 900 python bench.py --steps 500 --warmup 50 --no-cpu-baseline --no-config-c > $O/bench_sharded_world1_fixed.json 2> /dev/null
ls -la $O
