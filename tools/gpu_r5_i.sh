#!/bin/bash
O=gpurun_out/r5i; mkdir -p $O
B="--no-cpu-baseline --no-cache-tier --no-laia --no-wide --no-cold-tier"
for sp in 16 0; do
  timeout 300 python bench.py --span $sp $B > $O/b_long_span$sp.json 2> $O/b_long_span$sp.err
  timeout 300 python bench.py --steps 20 --warmup 5 --span $sp $B > $O/b_short_span$sp.json 2> $O/b_short_span$sp.err
done
HIP_FORCE_DEV_KERNARG=1 timeout 300 python bench.py --span 16 $B > $O/b_long_span16_dk.json 2> $O/b_long_span16_dk.err
HA_QSPAN_WAIT_COUNTS=0 timeout 300 python bench.py --span 16 $B > $O/b_long_span16_nowaitcounts.json 2> $O/b_long_span16_nowaitcounts.err
timeout 300 python bench.py --span 16 --queue-block 32 $B > $O/b_long_span16_block32.json 2> $O/b_long_span16_block32.err
timeout 300 python bench.py --span 32 --queue-block 32 $B > $O/b_long_span32_block32.json 2> $O/b_long_span32_block32.err
for f in $O/b_*.json; do echo $f; python -c "
import json,sys
try:
    d=json.loads(open('$f').read().strip().splitlines()[-1]); print(d['ms_per_step'], d['roofline']['frac'], d.get('enqueue_ms'), d.get('device_ms'), d.get('host_bound'))
except Exception as e: print('ERR', e)
"; done
tail -n 2 $O/*.err | grep -v amdgpu.ids
