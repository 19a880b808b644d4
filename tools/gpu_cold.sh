#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
B="--no-cpu-baseline --no-cache-tier --no-laia --no-cold-tier --no-wide"
show() { python -c "
import json,sys
try:
    d=json.loads(open('$1').read().strip().splitlines()[-1]); print('$2: us/step %.2f  dev_ms %.3f enq_ms %.3f frac %.3f' % (d['ms_per_step']*1e3, d['device_ms'], d['enqueue_ms'], d['roofline']['frac']))
except Exception as e: print('$2: no result', e)"; }
for rep in 1 2; do
timeout 300 python bench.py $B --steps 20 --warmup 5 --pre-roll 0 --no-gate > /tmp/c1.json 2>/dev/null; show /tmp/c1.json "20 steps, as round 2"
timeout 300 python bench.py $B --steps 20 --warmup 5 --pre-roll 0 > /tmp/c2.json 2>/dev/null; show /tmp/c2.json "20 steps, gate only"
timeout 300 python bench.py $B --steps 20 --warmup 5 --no-gate > /tmp/c3.json 2>/dev/null; show /tmp/c3.json "20 steps, pre-roll 2 only"
timeout 300 python bench.py $B --steps 20 --warmup 5 > /tmp/c4.json 2>/dev/null; show /tmp/c4.json "20 steps, pre-roll 2 + gate"
timeout 300 python bench.py $B --steps 20 --warmup 5 --pre-roll 4 > /tmp/c5.json 2>/dev/null; show /tmp/c5.json "20 steps, pre-roll 4 + gate"
done
timeout 300 python bench.py $B > /tmp/c6.json 2>/dev/null; show /tmp/c6.json "2048 steps default"
timeout 300 python bench.py $B --engine queue --queue-block 16 --steps 20 --warmup 5 > /tmp/c7.json 2>/dev/null; show /tmp/c7.json "queue block16 20 steps"
