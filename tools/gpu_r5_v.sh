#!/bin/bash
O=gpurun_out/r5v; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_fullscale.py -m gpu -x -q -k "wide_queue_step" -s > $O/t_wide_full.log 2>&1; tail -4 $O/t_wide_full.log | cut -c1-300
timeout 900 python -m pytest tests/test_gpu_example_wdl.py tests/test_gpu_laia.py tests/test_gpu_laia_config_d.py tests/test_gpu_hetu_ops.py -m gpu -x -q > $O/t_misc.log 2>&1; tail -3 $O/t_misc.log | cut -c1-300
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-cold-tier > $O/b_short.json 2> $O/b_short.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r5v/b_short.json').read().strip().splitlines()[-1])
print('headline', d['ms_per_step'], d['roofline']['frac'])
for k in ('cache_tier','laia_scheduler','wide_bs1024_d512','wide_bs4096_d128'):
    v=d.get(k); 
    if not v: continue
    print(k, {x: v[x] for x in v if x in ('us_per_step','us_per_global_batch','in_call_us_per_global_batch','error','thread_wall_us_per_global_batch')})
PY
