#!/bin/bash
# round 4, GPU session S: the preparation stream confined to a share of the compute units
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r4s; mkdir -p $O
for sh in "4096 128" "1024 512"; do set -- $sh
  for cus in 0 2 3 4 8; do
    HA_QSIDE_CUS=$cus BATCH=$1 WIDTH=$2 timeout 600 python tools/shape_bench.py 2>/dev/null | head -1 | sed "s/^/side CUs 1 in $cus: /" >> $O/shapes.txt
  done
done
B="--no-cpu-baseline --no-cache-tier --no-laia --no-cold-tier --no-wide"
for cus in 0 4 8 16; do
HA_QSIDE_CUS=$cus timeout 400 python bench.py $B 2>/dev/null | python tools/ab_line.py "side CUs 1 in $cus" long >> $O/ab.txt
HA_QSIDE_CUS=$cus timeout 400 python bench.py $B --steps 20 --warmup 5 2>/dev/null | python tools/ab_line.py "side CUs 1 in $cus" short >> $O/ab.txt
done
cat $O/shapes.txt $O/ab.txt
