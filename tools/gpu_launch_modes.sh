#!/bin/bash
# the work-queue step with plain launches and with hipGraph replays, long run and the 20-step run (same box)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
B="--no-cpu-baseline --no-cache-tier --no-laia --no-cold-tier --no-wide"
show() { python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1: us/step %.2f  dev_ms %.3f enq_ms %.3f host_bound %s' % (d['ms_per_step']*1e3, d['device_ms'], d['enqueue_ms'], d['host_bound']))"; }
for rep in 1 2; do
timeout 300 python bench.py $B --graph-steps 16 2>/dev/null | show "graphs, gate"
timeout 300 python bench.py $B --no-gate 2>/dev/null | show "eager, no gate"
timeout 300 python bench.py $B --no-gate --steps 20 --warmup 5 2>/dev/null | show "eager, no gate, 20/5"
timeout 300 python bench.py $B --graph-steps 16 --steps 20 --warmup 5 2>/dev/null | show "graphs, gate, 20/5"
done
