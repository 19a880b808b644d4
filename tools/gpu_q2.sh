#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
B="--no-cpu-baseline --no-cache-tier --no-laia --no-cold-tier"
show() { python -c "
import json,sys
try:
    d=json.loads(open('$1').read().strip().splitlines()[-1]); print('$2: us/step %.2f  dev_ms %.3f enq_ms %.3f' % (d['ms_per_step']*1e3, d['device_ms'], d['enqueue_ms']))
except Exception as e: print('$2: no result', e)"; }
for k in 2 3 4; do
timeout 600 python bench.py $B --engine queue --queue-side-streams $k --distinct-batches 1056 > /tmp/b3.json 2>/tmp/b3.err; show /tmp/b3.json "overlap, $k side streams"; tail -2 /tmp/b3.err
done
