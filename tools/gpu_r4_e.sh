#!/bin/bash
# round 4, GPU session E: wide path (tests + shapes), same-box A/B of the launch geometry, contract / cache / example tests
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4e; mkdir -p $O
timeout 2400 python -m pytest tests/test_gpu_qstep.py -x -q -m gpu > $O/t_qstep.log 2>&1; echo "qstep rc $?" >> $O/rc.txt
timeout 1500 python -m pytest tests/test_gpu_example_wdl.py tests/test_gpu_bench_contract.py tests/test_gpu_cache.py tests/test_plugins.py tests/test_gpu_laia.py tests/test_gpu_laia_config_d.py -x -q -m gpu > $O/t_misc.log 2>&1; echo "example+contract+cache rc $?" >> $O/rc.txt
for sh in "4096 128" "1024 512"; do set -- $sh
  for sy in flags events; do
    BATCH=$1 WIDTH=$2 SYNC=$sy timeout 600 python tools/shape_bench.py 2>/dev/null | head -1 >> $O/shapes.txt
  done
  BATCH=$1 WIDTH=$2 BLOCK=8 timeout 600 python tools/shape_bench.py 2>/dev/null | head -1 | sed 's/$/ block8/' >> $O/shapes.txt
done
REPS=2 timeout 2400 bash tools/ab_variants.sh "wg256:" "gold_r3geometry:-DQV_GOLD=1 -DQV_WG=1024 -DQV_COOPSLOTS=64" "wg512:-DQV_WG=512 -DQV_COOPSLOTS=128" "wg256_rowldnt:-DQV_ROWLD_NT=1" > $O/variants.txt 2>&1
timeout 400 python tools/framed_hostprof.py 2>&1 | grep "us/step" > $O/framed_hostprof.txt
ls -la $O
