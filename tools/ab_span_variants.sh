#!/bin/bash
# Measurement-only build variants of the spanning launch on ONE box: the per-wave timeline of each (tools/qspan_timeline.py on
# a table of ROWS rows).  usage: ab_span_variants.sh "<name>:<-D flags>" ...  (through gpurun; the box copy's library is relinked)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=${OUT:-gpurun_out/ab_span}; mkdir -p $O
FL="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-gpu-rdc -I include"
OBJS=$(ls herald_amd/_build/*.o | grep -v qstep.o)
for v in "$@"; do
  name=${v%%:*}; flags=${v#*:}
  /opt/rocm/bin/hipcc $FL $flags -c herald_amd/csrc/qstep.hip -o /tmp/qstep_$name.o || { echo "$name: compile failed"; continue; }
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -fno-gpu-rdc -o herald_amd/libherald_amd.so $OBJS /tmp/qstep_$name.o || continue
  ROWS=${ROWS:-8000000} timeout 300 python tools/qspan_timeline.py > $O/timeline_$name.txt 2>&1
  echo "== $name ($flags)"; grep -E "^launch of|^step  0|^step  8|^step 12|S apply|S copy|^  [GLMS] " $O/timeline_$name.txt
done
