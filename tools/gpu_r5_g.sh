#!/bin/bash
O=gpurun_out/r5g; mkdir -p $O
timeout 300 python tools/span_debug.py 16 16 flags 2 > $O/span_debug.log 2>&1; tail -3 $O/span_debug.log
ROWS=8000000 timeout 300 python tools/qspan_timeline.py > $O/timeline_base.txt 2>&1
HIP_FORCE_DEV_KERNARG=1 ROWS=8000000 timeout 300 python tools/qspan_timeline.py > $O/timeline_devkernarg.txt 2>&1
for n in base devkernarg; do echo "== $n"; grep -E "^launch of|^step  0|^step  1:|^step  8|^step 12|S apply|S copy|^  [GLMS] " $O/timeline_$n.txt; done
