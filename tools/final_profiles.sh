#!/bin/bash
# One gpurun call that regenerates everything kept under profiles/<tag> (run through gpurun from the repo root).
TAG=${1:-r02}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/final_$TAG
mkdir -p $O
bash $R/tools/profile_round.sh $TAG > $O/profile_round.log 2>&1
cp $R/gpurun_out/prof_$TAG/summary/bench_kernel_stats.csv $R/gpurun_out/prof_$TAG/summary/pmc_traffic.json $O/ 2>/dev/null
cd $R
# bench lines: default, the driver's short run, and under rocprof (the run the kernel stats come from)
python3 bench.py > $O/bench_n1_default.json 2> $O/bench_n1_default.err
python3 bench.py --steps 20 --warmup 5 > $O/bench_n1_steps20_warmup5.json 2> /dev/null
grep '^{' gpurun_out/prof_$TAG/stats.log | tail -1 > $O/bench_under_rocprof.json
python3 bench.py --lookahead 3 --no-cpu-baseline --no-cache-tier --no-cold-tier --no-laia --no-wide > $O/bench_n1_lookahead3.json 2> /dev/null
# shapes
for bs in 512 1024 4096; do w=128; [ $bs = 1024 ] && w=512
  BATCH=$bs WIDTH=$w python3 tools/cfgc_bench.py 2>&1 | grep -v amdgpu.ids | tail -13 > $O/shape_bs${bs}_d${w}.txt
done
# world-size-1 sharded step
MASTER_ADDR=127.0.0.1 MASTER_PORT=29633 RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 HA_FORCE_SHARDED=1 python3 bench.py --steps 500 --warmup 50 --no-cpu-baseline 2>/dev/null | grep '^{' > $O/bench_sharded_world1.json
python3 tools/shard_hostprof.py 2>&1 | grep "us/step" > $O/sharded_world1_step.txt
ls -la $O
