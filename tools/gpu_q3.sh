#!/bin/bash
# quick A/B of a qapply change: tests, default bench (long + 20-step), timeline
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/q3; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_qstep.py -q -x --timeout 600 2>&1 | tail -3
B="--no-cpu-baseline --no-cache-tier --no-laia --no-cold-tier"
for extra in "" "--steps 20 --warmup 5" "" "--steps 20 --warmup 5"; do
timeout 600 python bench.py $B $extra 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$extra', 'us/step %.2f frac %.3f' % (d['ms_per_step']*1e3, d['roofline']['frac']))"
done
timeout 600 python tools/qstep_timeline.py 2>&1 | grep -v amdgpu.ids | head -14
