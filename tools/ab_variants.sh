#!/bin/bash
# A/B of compile-time variants of csrc/qstep.hip on ONE box (box-to-box differences are larger than the effects looked
# for): usage  ab_variants.sh "<name>:<-D flags>" ...   (run through gpurun; the library of the box copy is relinked)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
B="--no-cpu-baseline --no-cache-tier --no-laia --no-cold-tier --no-wide"
FL="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-gpu-rdc -I include"
OBJS=$(ls herald_amd/_build/*.o | grep -v "/qstep.o")
REPS=${REPS:-2}
line() { python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-28s %s us/step %.2f' % ('$1', '$2', d['ms_per_step']*1e3))"; }
for rep in $(seq $REPS); do
for v in "$@"; do
  name=${v%%:*}; flags=${v#*:}
  /opt/rocm/bin/hipcc $FL $flags -c herald_amd/csrc/qstep.hip -o /tmp/qstep_$name.o || { echo "$name: compile failed"; continue; }
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -fno-gpu-rdc -o herald_amd/libherald_amd.so $OBJS /tmp/qstep_$name.o || continue
  timeout 300 python bench.py $B $BENCH_EXTRA 2>/dev/null | line "$name" long
  timeout 300 python bench.py $B --steps 20 --warmup 5 $BENCH_EXTRA 2>/dev/null | line "$name" short
done
done
