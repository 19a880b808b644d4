#!/bin/bash
# One gpurun call that regenerates what is kept under profiles/r06 from the final binaries (run from the repo root):
#   bench lines (default, the driver's short run, under rocprofv3), kernel stats, PMC traffic (headline and both wide shapes),
#   the per-wave timeline + census of one apply launch, the cache tier's planned flow by kernel (three policies), the sharded
#   step at world size 1
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/final_r06
mkdir -p $O/summary $R/profiles/r06
cd $R
B="--no-cache-tier --no-laia --no-cold-tier --no-wide --no-sweeps"
# ---- N=1: kernel trace + stats, PMC traffic (separate passes) ----
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 512 --warmup 64 --no-cpu-baseline --no-kernel-pass $B"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $R/bench.py $ARGS > $O/stats.log 2>&1
PA="--steps 64 --warmup 32 --no-cpu-baseline --no-kernel-pass $B"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- python3 $R/bench.py $PA > $O/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- python3 $R/bench.py $PA > $O/pmc_write.log 2>&1
python3 $R/tools/pmc_summary.py $O $O/summary > $O/pmc_summary.log 2>&1
grep '^{' $O/stats.log | tail -1 > $O/summary/bench_under_rocprof.json
cp $O/summary/pmc_traffic.json $R/profiles/r06/pmc_traffic.json 2>/dev/null     # bench.py reads the newest profiles/r*/pmc_traffic.json
# ---- the wide shapes: PMC traffic per shape (tools/shape_bench.py is the measurement bench.py reports as wide_*) ----
for sh in "1024 512" "4096 128"; do set -- $sh
  mkdir -p $O/wide_$1
  BATCH=$1 WIDTH=$2 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/wide_$1/pmc_fetch -- python3 $R/tools/shape_bench.py > $O/wide_$1/pmc_fetch.log 2>&1
  BATCH=$1 WIDTH=$2 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/wide_$1/pmc_write -- python3 $R/tools/shape_bench.py > $O/wide_$1/pmc_write.log 2>&1
  python3 $R/tools/pmc_summary.py $O/wide_$1 $O/wide_$1/summary > /dev/null 2>&1
  cp $O/wide_$1/summary/pmc_traffic.json $O/summary/pmc_traffic_wide_bs$1_d$2.json 2>/dev/null
  cp $O/wide_$1/summary/pmc_traffic.json $R/profiles/r06/pmc_traffic_wide_bs$1_d$2.json 2>/dev/null
done
# ---- the cache tier's planned flow by kernel, per policy ----
for pol in LRU LFU LFUOpt; do
  POLICY=$pol BLOCKS=8 rocprofv3 --kernel-trace --stats --output-format csv -d $O/cache_$pol -- python3 $R/tools/cache_planned_bench.py > $O/cache_$pol.log 2>&1
  ( grep "planned:" $O/cache_$pol.log; KTRACE=$O/cache_$pol python3 $R/tools/cache_planned_bench.py ) > $O/summary/cache_planned_$pol.txt 2>&1
done
cd $R
# ---- bench lines on the final binaries ----
python3 bench.py > $O/summary/bench_n1_default.json 2> $O/bench_n1_default.err
python3 bench.py --steps 20 --warmup 5 > $O/summary/bench_n1_steps20_warmup5.json 2> /dev/null
for i in 1 2 3; do python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline $B 2>/dev/null | python3 tools/ab_line.py driver_run short >> $O/summary/short_runs.txt; done
# ---- timeline + census of one apply launch; the wide path's step alone / beside its preparation ----
python3 tools/qstep_timeline.py 2>&1 | grep -v amdgpu > $O/summary/timeline_qapply_census.txt
for sh in "4096 128" "1024 512"; do set -- $sh
  ALONE=1 BATCH=$1 WIDTH=$2 python3 tools/shape_bench.py 2>/dev/null > $O/summary/shape_bs$1_d$2.txt
done
# ---- the laia scheduler's global batch by kernel ----
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/laia_stats -o laia -- python3 $R/tools/laia_profile.py > $O/laia_stats.log 2>&1
f=$(find $O/laia_stats -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/summary/laia_kernel_stats.csv
python3 $R/tools/laia_gaps.py $O/laia_stats > $O/summary/laia_gaps.txt 2>/dev/null
cd $R
# ---- the sharded step at world size 1 (the N>1 code path on one GPU) ----
export MASTER_ADDR=127.0.0.1 MASTER_PORT=29633 RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 HA_FORCE_SHARDED=1
python3 bench.py --steps 500 --warmup 60 --no-cpu-baseline 2>/dev/null | grep '^{' > $O/summary/bench_sharded_world1.json
unset HA_FORCE_SHARDED RANK WORLD_SIZE LOCAL_RANK
find $O -name "*.csv" -size +3M -delete
ls -la $O/summary
