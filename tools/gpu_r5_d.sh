#!/bin/bash
# the 256-thread geometry in span mode: where does the time go?
O=gpurun_out/r5d; mkdir -p $O
FL="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-gpu-rdc -I include"
OBJS=$(ls herald_amd/_build/*.o | grep -v qstep.o)
/opt/rocm/bin/hipcc $FL -DQV_GOLD=0 -DQV_WG=256 -c herald_amd/csrc/qstep.hip -o /tmp/qstep_256.o || exit 1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -fno-gpu-rdc -o herald_amd/libherald_amd.so $OBJS /tmp/qstep_256.o || exit 1
WPW=4 timeout 300 python tools/qspan_timeline.py > $O/timeline_span_wg256.txt 2>&1
head -30 $O/timeline_span_wg256.txt; tail -25 $O/timeline_span_wg256.txt
