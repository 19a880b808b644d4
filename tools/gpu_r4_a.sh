#!/bin/bash
# round 4, GPU session A: new parity tests, today's baseline, yardsticks, timeline, wait counters
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4a; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_qstep.py -x -q -m gpu -k "medium_items or criteo_stream" -s > $O/t_qstep.log 2>&1; echo "qstep rc $?" >> $O/rc.txt
timeout 1500 python -m pytest tests/test_gpu_fullscale.py -x -q -m gpu -k "queue_step" -s > $O/t_full.log 2>&1; echo "fullscale rc $?" >> $O/rc.txt
B="--no-cpu-baseline --no-cache-tier --no-laia --no-cold-tier --no-wide"
timeout 400 python bench.py $B > $O/bench_long.json 2> $O/bench_long.err
timeout 400 python bench.py $B --steps 20 --warmup 5 > $O/bench_short.json 2>/dev/null
timeout 300 tools/_bin/floor_bench 16 > $O/floor.txt 2>&1
timeout 400 python tools/qstep_timeline.py > $O/timeline.txt 2>&1
timeout 2400 bash tools/pmc_wait_counters.sh r04 > $O/pmcw.log 2>&1
cp gpurun_out/pmcw_r04/summary.json gpurun_out/pmcw_r04/groups.txt $O/ 2>/dev/null
grep -c . gpurun_out/pmcw_r04/counters_available.txt >> $O/rc.txt
# keep the merged-back size small: drop the raw per-dispatch csv
find gpurun_out/pmcw_r04 -name "*.csv" -size +2M -delete
ls -la $O
