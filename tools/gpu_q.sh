#!/bin/bash
# quick GPU pass for the work-queue step: tests (both ranking forms), timeline, bench
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/${1:-q}; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_qstep.py -q -x --timeout 300 > $O/qstep_tests.log 2>&1; echo "qstep tests rc=$?"
tail -3 $O/qstep_tests.log
HA_QSTEP_BALLOT=1 timeout 900 python -m pytest tests/test_gpu_qstep.py -q -x --timeout 300 -k "small_tables or ragged" > $O/qstep_tests_ballot.log 2>&1; echo "qstep tests (ballot form) rc=$?"
tail -2 $O/qstep_tests_ballot.log
B="--no-cpu-baseline --no-cache-tier --no-laia --no-cold-tier"
for e in queue "queue --queue-block 4" "queue --queue-block 16" handoff; do
  timeout 600 python bench.py $B --engine $e > "$O/bench_$e.json" 2> "$O/bench_$e.err"; echo "bench $e rc=$?"
  python - <<PY
import json
try:
    d=json.loads(open("$O/bench_$e.json").read().strip().splitlines()[-1])
    print("$e", "us/step %.2f" % (d["ms_per_step"]*1e3), "frac %.3f" % d["roofline"]["frac"], "rows/s %.1fM" % (d["value"]/1e6))
except Exception as ex:
    print("$e: no result", ex)
PY
done
timeout 600 python bench.py $B --engine queue --steps 20 --warmup 5 > $O/bench_queue_20.json 2>> $O/bench_queue.err
python -c "
import json; d=json.loads(open('$O/bench_queue_20.json').read().strip().splitlines()[-1]); print('queue 20 steps: us/step %.2f' % (d['ms_per_step']*1e3))"
