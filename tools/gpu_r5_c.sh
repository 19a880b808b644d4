#!/bin/bash
O=gpurun_out/r5c; mkdir -p $O
REPS=1 BENCH_EXTRA="--span 16" bash tools/ab_variants.sh "wg1024:" "wg512:-DQV_GOLD=0 -DQV_WG=512" "wg256:-DQV_GOLD=0 -DQV_WG=256" > $O/ab_span16.txt 2>&1
REPS=1 BENCH_EXTRA="--span 0" bash tools/ab_variants.sh "wg1024_span0:" >> $O/ab_span16.txt 2>&1
cat $O/ab_span16.txt
