#!/bin/bash
# round 4, GPU session G: chunked workgroup items: parity, shapes, headline sanity
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4g; mkdir -p $O
timeout 2400 python -m pytest tests/test_gpu_qstep.py tests/test_gpu_tolerance.py -x -q -m gpu > $O/t_qstep.log 2>&1; echo "qstep rc $?" >> $O/rc.txt
timeout 1500 python -m pytest tests/test_gpu_fullscale.py -x -q -m gpu -k "queue_step" > $O/t_full.log 2>&1; echo "fullscale rc $?" >> $O/rc.txt
for sh in "4096 128" "1024 512" "2048 128"; do set -- $sh
  BATCH=$1 WIDTH=$2 timeout 600 python tools/shape_bench.py 2>/dev/null > $O/shape_$1_$2.txt; head -1 $O/shape_$1_$2.txt >> $O/shapes.txt
done
B="--no-cpu-baseline --no-cache-tier --no-laia --no-cold-tier --no-wide"
timeout 400 python bench.py $B 2>/dev/null | python tools/ab_line.py final long >> $O/ab.txt
timeout 400 python bench.py $B --steps 20 --warmup 5 2>/dev/null | python tools/ab_line.py final short >> $O/ab.txt
timeout 600 python -m pytest tests/test_gpu_bench_contract.py -x -q -m gpu > $O/t_contract.log 2>&1; echo "contract rc $?" >> $O/rc.txt
ls -la $O
