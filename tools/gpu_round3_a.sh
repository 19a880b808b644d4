#!/bin/bash
# first GPU pass of round 3: the work-queue step (tests, bench next to the other engines, timeline), then the new parity tests
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r03a; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_qstep.py -q -x --timeout 300 > $O/qstep_tests.log 2>&1; echo "qstep tests rc=$?" | tee -a $O/summary.txt
tail -30 $O/qstep_tests.log
B="--no-cpu-baseline --no-cache-tier --no-laia --no-cold-tier"
for e in queue handoff forward; do
  timeout 600 python bench.py $B --engine $e > $O/bench_$e.json 2> $O/bench_$e.err; echo "bench $e rc=$?" | tee -a $O/summary.txt
  python - <<PY
import json
try:
    d=json.loads(open("$O/bench_$e.json").read().strip().splitlines()[-1])
    print("$e", "us/step %.2f" % (d["ms_per_step"]*1e3), "frac %.3f" % d["roofline"]["frac"], "rows/s %.1fM" % (d["value"]/1e6))
except Exception as ex:
    print("$e: no result", ex)
PY
done | tee -a $O/summary.txt
timeout 600 python bench.py $B --engine queue --steps 20 --warmup 5 > $O/bench_queue_20.json 2>> $O/bench_queue.err
timeout 600 python tools/qstep_timeline.py > $O/qstep_timeline.txt 2>&1; echo "timeline rc=$?" | tee -a $O/summary.txt
tail -40 $O/qstep_timeline.txt
timeout 900 python -m pytest tests/test_gpu_golden.py tests/test_gpu_cache.py -q --timeout 300 > $O/golden_tests.log 2>&1; echo "golden+cache tests rc=$?" | tee -a $O/summary.txt
tail -5 $O/golden_tests.log
timeout 1500 python -m pytest tests/test_gpu_fullscale.py -q --timeout 900 -s > $O/fullscale_tests.log 2>&1; echo "fullscale tests rc=$?" | tee -a $O/summary.txt
tail -15 $O/fullscale_tests.log
cat $O/summary.txt
