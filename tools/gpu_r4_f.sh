#!/bin/bash
# round 4, GPU session F: 1024-thread geometry back as the default + half-wave pairs for narrow rows: parity, shapes, A/B
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4f; mkdir -p $O
timeout 2400 python -m pytest tests/test_gpu_qstep.py tests/test_gpu_tolerance.py -x -q -m gpu > $O/t_qstep.log 2>&1; echo "qstep rc $?" >> $O/rc.txt
timeout 1500 python -m pytest tests/test_gpu_fullscale.py -x -q -m gpu -k "queue_step" > $O/t_full.log 2>&1; echo "fullscale rc $?" >> $O/rc.txt
for sh in "4096 128" "1024 512" "4096 64" "2048 128"; do set -- $sh
  BATCH=$1 WIDTH=$2 timeout 600 python tools/shape_bench.py 2>/dev/null | head -1 >> $O/shapes.txt
done
REPS=2 timeout 2400 bash tools/ab_variants.sh "gold1024:" "wg256:-DQV_GOLD=0" "wg512:-DQV_GOLD=0 -DQV_WG=512 -DQV_COOPSLOTS=128" > $O/variants.txt 2>&1
MASTER_ADDR=127.0.0.1 MASTER_PORT=29633 RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 HA_FORCE_SHARDED=1 timeout 900 python bench.py --steps 500 --warmup 50 --no-cpu-baseline 2>/dev/null | grep '^{' > $O/bench_sharded_world1.json
timeout 1200 python -m pytest tests/test_gpu_framed.py tests/test_gpu_example_wdl.py -x -q -m gpu > $O/t_framed.log 2>&1; echo "framed+example rc $?" >> $O/rc.txt
ls -la $O
