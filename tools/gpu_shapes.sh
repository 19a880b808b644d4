#!/bin/bash
# the per-GPU shapes of BASELINE configs[2] / [3] with and without the tolerance mode
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/shapes; mkdir -p $O
for tol in 0 1; do
for bs in 1024 4096; do w=128; [ $bs = 1024 ] && w=512
  TOL=$tol BATCH=$bs WIDTH=$w timeout 600 python3 tools/cfgc_bench.py 2>&1 | grep -v amdgpu.ids | tail -14 > $O/shape_bs${bs}_d${w}_tol$tol.txt
  echo "== bs=$bs d=$w tol=$tol"; cat $O/shape_bs${bs}_d${w}_tol$tol.txt
done; done
