#!/bin/bash
# round 5, first look at the spanning launch: parity tests, then same-box A/B against one launch per step
O=gpurun_out/r5a; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_qstep.py -m gpu -x -q -k "qspan" > $O/t_qspan.log 2>&1; echo "qspan rc=$?" >> $O/t_qspan.log
tail -5 $O/t_qspan.log
for sp in 0 16; do
  timeout 300 python bench.py --steps 20 --warmup 5 --span $sp --no-cpu-baseline --no-cache-tier --no-laia --no-wide --no-cold-tier > $O/b_short_span$sp.json 2> $O/b_short_span$sp.err
  timeout 300 python bench.py --span $sp --no-cpu-baseline --no-cache-tier --no-laia --no-wide --no-cold-tier > $O/b_long_span$sp.json 2> $O/b_long_span$sp.err
done
for f in $O/b_*.json; do echo $f; python -c "
import json,sys
try:
    d=json.loads(open('$f').read().strip().splitlines()[-1]); print(d['ms_per_step'], d['roofline']['frac'], d.get('enqueue_ms'), d.get('device_ms'))
except Exception as e: print('ERR', e)
"; done
tail -3 $O/*.err
