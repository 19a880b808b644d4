#!/bin/bash
# round 4, GPU session I: per-kernel durations of the wide shapes (rocprofv3 stats), BLOCK sweep, stream-priority A/B
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r4i; mkdir -p $O
export TMPDIR=/tmp
for sh in "4096 128" "1024 512"; do set -- $sh
  export BATCH=$1 WIDTH=$2
  timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof_$1_$2 -o s -- python3 tools/shape_bench.py > $O/prof_$1_$2.log 2>&1
  f=$(find $O/prof_$1_$2 -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/kstats_$1_$2.csv
  find $O/prof_$1_$2 -name "*kernel_trace.csv" -size +60M -delete
  for blk in 2 8; do
    BLOCK=$blk timeout 600 python tools/shape_bench.py 2>/dev/null | head -1 | sed "s/^/block $blk: /" >> $O/blocks.txt
  done
done
unset BATCH WIDTH
B="--no-cpu-baseline --no-cache-tier --no-laia --no-cold-tier --no-wide"
for i in 1 2; do
timeout 400 python bench.py $B 2>/dev/null | python tools/ab_line.py base long >> $O/prio.txt
timeout 400 python bench.py $B --steps 20 --warmup 5 2>/dev/null | python tools/ab_line.py base short >> $O/prio.txt
HA_QSIDE_PRIO=low HA_BENCH_MAIN_PRIO=high timeout 400 python bench.py $B 2>/dev/null | python tools/ab_line.py prio long >> $O/prio.txt
HA_QSIDE_PRIO=low HA_BENCH_MAIN_PRIO=high timeout 400 python bench.py $B --steps 20 --warmup 5 2>/dev/null | python tools/ab_line.py prio short >> $O/prio.txt
done
ls -la $O; cat $O/blocks.txt $O/prio.txt; head -12 $O/kstats_4096_128.csv
