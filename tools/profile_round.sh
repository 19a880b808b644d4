#!/bin/bash
# Collects the rocprofv3 evidence for one round on the GPU box (run through gpurun):
#   gpurun_out/prof_<tag>/stats   kernel trace + stats of the default bench command
#   gpurun_out/prof_<tag>/pmc_*   FETCH_SIZE / WRITE_SIZE passes (separate runs, as the guide prescribes)
TAG=${1:-r02}
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 512 --warmup 64 --no-cpu-baseline --no-kernel-pass --no-cache-tier --no-cold-tier --no-laia --no-wide"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $GRAFT_REPO_ROOT/bench.py $ARGS > $OUT/stats.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $GRAFT_REPO_ROOT/bench.py --steps 64 --warmup 32 --graph-steps 1 --no-cpu-baseline --no-kernel-pass --no-cache-tier --no-cold-tier --no-laia --no-wide > $OUT/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 $GRAFT_REPO_ROOT/bench.py --steps 64 --warmup 32 --graph-steps 1 --no-cpu-baseline --no-kernel-pass --no-cache-tier --no-cold-tier --no-laia --no-wide > $OUT/pmc_write.log 2>&1
ls $OUT
python3 $GRAFT_REPO_ROOT/tools/pmc_summary.py $OUT $OUT/summary
