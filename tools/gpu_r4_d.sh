#!/bin/bash
# round 4, GPU session D: epoch/flag stream ordering, the wide path, workgroup-size A/B on one box
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4d; mkdir -p $O
timeout 2400 python -m pytest tests/test_gpu_qstep.py -x -q -m gpu > $O/t_qstep.log 2>&1; echo "qstep rc $?" >> $O/rc.txt
timeout 1500 python -m pytest tests/test_gpu_fullscale.py -x -q -m gpu -k "queue_step" -s > $O/t_full.log 2>&1; echo "fullscale rc $?" >> $O/rc.txt
timeout 1500 python -m pytest tests/test_gpu_example_wdl.py tests/test_gpu_framed.py tests/test_gpu_bench_contract.py -x -q -m gpu > $O/t_example.log 2>&1; echo "example+framed rc $?" >> $O/rc.txt
B="--no-cpu-baseline --no-cache-tier --no-laia --no-cold-tier --no-wide"
for i in 1 2; do
timeout 400 python bench.py $B 2>/dev/null | python tools/ab_line.py flags long >> $O/ab.txt
timeout 400 python bench.py $B --steps 20 --warmup 5 2>/dev/null | python tools/ab_line.py flags short >> $O/ab.txt
HA_QSYNC=events timeout 400 python bench.py $B 2>/dev/null | python tools/ab_line.py events long >> $O/ab.txt
HA_QSYNC=events timeout 400 python bench.py $B --steps 20 --warmup 5 2>/dev/null | python tools/ab_line.py events short >> $O/ab.txt
done
timeout 400 python tools/qstep_timeline.py > $O/timeline.txt 2>&1
REPS=2 timeout 2400 bash tools/ab_variants.sh "wg256:" "wg1024:-DQV_WG=1024 -DQV_COOPSLOTS=64" "wg512:-DQV_WG=512 -DQV_COOPSLOTS=128" "wg1024nt:-DQV_WG=1024 -DQV_COOPSLOTS=64 -DQV_ROWLD_NT=1" > $O/variants.txt 2>&1
MASTER_ADDR=127.0.0.1 MASTER_PORT=29633 RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 HA_FORCE_SHARDED=1 timeout 900 python bench.py --steps 500 --warmup 50 --no-cpu-baseline --no-config-c 2>/dev/null | grep '^{' > $O/bench_sharded_world1.json
ls -la $O
