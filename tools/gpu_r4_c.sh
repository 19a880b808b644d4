#!/bin/bash
# round 4, GPU session C: the 256-thread apply launch -- parity, bench, A/B knobs, timeline; laia example; framed host profile
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4c; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_qstep.py tests/test_gpu_tolerance.py -x -q -m gpu > $O/t_qstep.log 2>&1; echo "qstep rc $?" >> $O/rc.txt
timeout 1500 python -m pytest tests/test_gpu_fullscale.py -x -q -m gpu -k "queue_step" -s > $O/t_full.log 2>&1; echo "fullscale rc $?" >> $O/rc.txt
timeout 1500 python -m pytest tests/test_gpu_example_wdl.py -x -q -m gpu > $O/t_example.log 2>&1; echo "example rc $?" >> $O/rc.txt
B="--no-cpu-baseline --no-cache-tier --no-laia --no-cold-tier --no-wide"
for i in 1 2; do
timeout 400 python bench.py $B 2>/dev/null | python tools/ab_line.py base long >> $O/ab.txt
timeout 400 python bench.py $B --steps 20 --warmup 5 2>/dev/null | python tools/ab_line.py base short >> $O/ab.txt
HA_QNOSYNC=1 timeout 400 python bench.py $B 2>/dev/null | python tools/ab_line.py nosync long >> $O/ab.txt
HA_QNOSYNC=1 timeout 400 python bench.py $B --steps 20 --warmup 5 2>/dev/null | python tools/ab_line.py nosync short >> $O/ab.txt
done
timeout 400 python tools/qstep_timeline.py > $O/timeline.txt 2>&1
timeout 400 python tools/framed_hostprof.py > $O/framed_hostprof.txt 2>&1
REPS=2 timeout 2400 bash tools/ab_variants.sh "base:" "rowldnt:-DQV_ROWLD_NT=1" "coop128:-DQV_COOPSLOTS=128" "coop512:-DQV_COOPSLOTS=512" "rowldnt_gradnt:-DQV_ROWLD_NT=1 -DQV_GRAD_NT=1" "outplain:-DQV_OUT_NT=0" > $O/variants.txt 2>&1
ls -la $O
