#!/bin/bash
O=gpurun_out/r5e; mkdir -p $O
timeout 300 python tools/span_debug.py 16 16 flags 3 > $O/span_debug.log 2>&1; tail -5 $O/span_debug.log
timeout 900 python -m pytest tests/test_gpu_qstep.py -m gpu -x -q -k "qspan" > $O/t_qspan.log 2>&1; tail -4 $O/t_qspan.log
timeout 300 python tools/qspan_timeline.py > $O/timeline_span.txt 2>&1; head -22 $O/timeline_span.txt
for sp in 16 0; do
  timeout 300 python bench.py --span $sp --no-cpu-baseline --no-cache-tier --no-laia --no-wide --no-cold-tier > $O/b_long_span$sp.json 2> $O/b_long_span$sp.err
  timeout 300 python bench.py --steps 20 --warmup 5 --span $sp --no-cpu-baseline --no-cache-tier --no-laia --no-wide --no-cold-tier > $O/b_short_span$sp.json 2> $O/b_short_span$sp.err
done
for f in $O/b_*.json; do echo $f; python -c "
import json,sys
try:
    d=json.loads(open('$f').read().strip().splitlines()[-1]); print(d['ms_per_step'], d['roofline']['frac'], d.get('enqueue_ms'), d.get('device_ms'))
except Exception as e: print('ERR', e)
"; done
tail -n 3 $O/*.err
