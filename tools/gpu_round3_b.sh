#!/bin/bash
# second GPU pass of round 3: the queue engine as the default -- tests, smoke, bench lines, timeline, rocprof evidence
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
R=$(pwd); O=$R/gpurun_out/r03b; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_qstep.py tests/test_gpu_bench_contract.py -q -x --timeout 600 > $O/tests.log 2>&1; echo "qstep+contract tests rc=$?" | tee -a $O/summary.txt
tail -8 $O/tests.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?" | tee -a $O/summary.txt; tail -3 $O/smoke.log
timeout 900 python bench.py > $O/bench_n1_default.json 2> $O/bench_n1_default.err; echo "bench default rc=$?" | tee -a $O/summary.txt
timeout 600 python bench.py --steps 20 --warmup 5 > $O/bench_n1_steps20_warmup5.json 2>/dev/null
B="--no-cpu-baseline --no-cache-tier --no-laia --no-cold-tier"
timeout 600 python bench.py $B --engine handoff > $O/bench_n1_handoff.json 2>/dev/null
timeout 600 python bench.py $B --engine handoff --steps 20 --warmup 5 > $O/bench_n1_handoff_20.json 2>/dev/null
for f in bench_n1_default bench_n1_steps20_warmup5 bench_n1_handoff bench_n1_handoff_20; do python - <<PY
import json
try:
    d=json.loads(open("$O/$f.json").read().strip().splitlines()[-1])
    print("$f", "us/step %.2f" % (d["ms_per_step"]*1e3), "frac %.3f" % d["roofline"]["frac"], "rows/s %.1fM" % (d["value"]/1e6))
except Exception as ex:
    print("$f: no result", ex)
PY
done | tee -a $O/summary.txt
timeout 600 python tools/qstep_timeline.py > $O/qstep_timeline.txt 2>&1; echo "timeline rc=$?" | tee -a $O/summary.txt
tail -25 $O/qstep_timeline.txt
bash tools/profile_round.sh r03 > $O/profile_round.log 2>&1; echo "profile rc=$?" | tee -a $O/summary.txt
tail -20 $O/profile_round.log
cp gpurun_out/prof_r03/summary/* $O/ 2>/dev/null
grep '^{' gpurun_out/prof_r03/stats.log | tail -1 > $O/bench_under_rocprof.json
cat $O/summary.txt
