#!/bin/bash
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r4y; mkdir -p $O
B="--no-cpu-baseline --no-cache-tier --no-laia --no-cold-tier --no-wide"
for i in 1 2 3; do
HA_BENCH_TRACE=1 timeout 400 python bench.py $B --steps 20 --warmup 5 2> $O/trace_$i.txt | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print({k:d[k] for k in ('device_ms','enqueue_ms','host_bound','ms_per_step')})" >> $O/lines.txt
grep -E "before_chunk|step_chunk" $O/trace_$i.txt | tail -8 >> $O/lines.txt
done
cat $O/lines.txt
