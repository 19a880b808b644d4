#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
B="--no-cpu-baseline --no-cache-tier --no-laia --no-cold-tier"
for rep in 1 2 3; do for cw in 0 2 6; do
timeout 600 python bench.py $B --steps 20 --warmup 5 --clock-warm $cw 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('clock-warm $cw: us/step %.2f' % (d['ms_per_step']*1e3))"
done; done
timeout 600 python bench.py $B 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('long: us/step %.2f' % (d['ms_per_step']*1e3))"
