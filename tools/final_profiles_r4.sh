#!/bin/bash
# One gpurun call that regenerates what is kept under profiles/r04 from the final binaries (run from the repo root):
#   bench lines (default, driver's short run, under rocprofv3), kernel stats, PMC traffic, wait counters, timeline,
#   floor yardsticks, wide shapes, the sharded step at world size 1 (both shapes, per-shape PMC traffic)
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/final_r04
mkdir -p $O
cd $R
B="--no-cache-tier --no-laia --no-cold-tier --no-wide"
# ---- N=1: kernel trace + stats, PMC traffic (separate passes) ----
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 512 --warmup 64 --no-cpu-baseline --no-kernel-pass $B"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $R/bench.py $ARGS > $O/stats.log 2>&1
PA="--steps 64 --warmup 32 --no-cpu-baseline --no-kernel-pass $B"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- python3 $R/bench.py $PA > $O/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- python3 $R/bench.py $PA > $O/pmc_write.log 2>&1
python3 $R/tools/pmc_summary.py $O $O/summary > $O/pmc_summary.log 2>&1
grep '^{' $O/stats.log | tail -1 > $O/summary/bench_under_rocprof.json
cd $R
mkdir -p profiles/r04 && cp $O/summary/pmc_traffic.json profiles/r04/pmc_traffic.json 2>/dev/null     # bench.py reads the newest profiles/r*/pmc_traffic.json
# ---- bench lines on the final binaries ----
python3 bench.py > $O/summary/bench_n1_default.json 2> $O/bench_n1_default.err
python3 bench.py --steps 20 --warmup 5 > $O/summary/bench_n1_steps20_warmup5.json 2> /dev/null
for i in 1 2 3; do python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline $B 2>/dev/null | python3 tools/ab_line.py driver_run short >> $O/summary/short_runs.txt; done
HA_QSYNC=events python3 bench.py --no-cpu-baseline $B 2>/dev/null | python3 tools/ab_line.py events_sync long >> $O/summary/short_runs.txt
python3 bench.py --no-cpu-baseline $B 2>/dev/null | python3 tools/ab_line.py flags_sync long >> $O/summary/short_runs.txt
python3 bench.py --engine handoff --no-cpu-baseline $B 2>/dev/null | grep '^{' > $O/summary/bench_n1_handoff.json
# ---- timeline, yardsticks, wide shapes ----
python3 tools/qstep_timeline.py 2>&1 | grep -v amdgpu > $O/summary/timeline_qapply.txt
tools/_bin/floor_bench 16 > $O/summary/floor_bench.txt 2>&1
for sh in "4096 128" "1024 512"; do set -- $sh
  ALONE=1 BATCH=$1 WIDTH=$2 python3 tools/shape_bench.py 2>/dev/null > $O/summary/shape_bs$1_d$2.txt
done
# ---- the laia scheduler's global batch by kernel ----
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/laia_stats -o laia -- python3 $R/tools/laia_profile.py > $O/laia_stats.log 2>&1
f=$(find $O/laia_stats -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/summary/laia_kernel_stats.csv
cd $R
# ---- the sharded step at world size 1 (the N>1 code path on one GPU), per-shape PMC ----
export MASTER_ADDR=127.0.0.1 MASTER_PORT=29633 RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 HA_FORCE_SHARDED=1
cd /tmp
for sh in "256 512" "4096 128"; do set -- $sh
  S="--batch $1 --width $2 --steps 64 --warmup 60 --no-cpu-baseline --no-config-c"
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/sh_stats_$1 -- python3 $R/bench.py --batch $1 --width $2 --steps 300 --warmup 60 --no-cpu-baseline --no-config-c > $O/sh_stats_$1.log 2>&1
  mkdir -p $O/shp_$1
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/shp_$1/pmc_fetch -- python3 $R/bench.py $S > $O/shp_$1/pmc_fetch.log 2>&1
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/shp_$1/pmc_write -- python3 $R/bench.py $S > $O/shp_$1/pmc_write.log 2>&1
  python3 $R/tools/pmc_summary.py $O/shp_$1 $O/shp_$1/summary > /dev/null 2>&1
  cp $O/shp_$1/summary/pmc_traffic.json $O/summary/pmc_traffic_sharded_bs$1_d$2.json 2>/dev/null
  cp $O/shp_$1/summary/pmc_traffic.json $R/profiles/r04/pmc_traffic_sharded_bs$1_d$2.json 2>/dev/null
  python3 - <<PY
import csv, glob
rows=[]
for f in glob.glob("$O/sh_stats_$1/*/*_kernel_stats.csv"):
    rd=csv.DictReader(open(f)); fields=rd.fieldnames; rows+=[r for r in rd if "ha::" in r["Name"]]
if rows:
    w=csv.DictWriter(open("$O/summary/sharded_world1_bs$1_d$2_kernel_stats.csv","w",newline=""),fieldnames=fields); w.writeheader(); w.writerows(rows)
PY
done
cd $R
python3 bench.py --steps 500 --warmup 60 --no-cpu-baseline 2>/dev/null | grep '^{' > $O/summary/bench_sharded_world1.json
HA_SHARD_FIXED=1 python3 bench.py --steps 500 --warmup 60 --no-cpu-baseline --no-config-c 2>/dev/null | grep '^{' > $O/summary/bench_sharded_world1_fixed_frames.json
python3 tools/framed_hostprof.py 2>&1 | grep "us/step" > $O/summary/sharded_world1_host_vs_wall.txt
unset HA_FORCE_SHARDED RANK WORLD_SIZE LOCAL_RANK
# keep the merge-back small
find $O -name "*.csv" -size +3M -delete
ls -la $O/summary
