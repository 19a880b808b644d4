#!/bin/bash
# round 4, GPU session N: the short run with the last-level cache as a long run leaves it
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r4n; mkdir -p $O
B="--no-cpu-baseline --no-cache-tier --no-laia --no-cold-tier --no-wide"
for i in 1 2 3; do
timeout 400 python bench.py $B --steps 20 --warmup 5 2>/dev/null | python tools/ab_line.py "w5 cache-warm 96 (default)" short >> $O/short.txt
timeout 400 python bench.py $B --steps 20 --warmup 5 --cache-warm 0 2>/dev/null | python tools/ab_line.py "w5 cache-warm 0" short >> $O/short.txt
timeout 400 python bench.py $B --steps 20 --warmup 5 --cache-warm 256 2>/dev/null | python tools/ab_line.py "w5 cache-warm 256" short >> $O/short.txt
timeout 400 python bench.py $B --steps 20 --warmup 5 --cache-warm 96 --clock-warm 0 2>/dev/null | python tools/ab_line.py "w5 cache-warm 96 clock-warm 0" short >> $O/short.txt
HA_QSYNC=events timeout 400 python bench.py $B --steps 20 --warmup 5 2>/dev/null | python tools/ab_line.py "w5 cache-warm 96 events" short >> $O/short.txt
timeout 400 python bench.py $B --steps 20 --warmup 101 --pre-roll 96 --cache-warm 0 2>/dev/null | python tools/ab_line.py "w101 pre-roll 96" short >> $O/short.txt
done
timeout 400 python bench.py $B 2>/dev/null | python tools/ab_line.py "default" long >> $O/short.txt
cat $O/short.txt
