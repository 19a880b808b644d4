#!/bin/bash
O=gpurun_out/r5b; mkdir -p $O
timeout 300 python tools/span_debug.py 16 16 flags 4 > $O/span_debug.log 2>&1; tail -6 $O/span_debug.log
timeout 300 python tools/qspan_timeline.py > $O/timeline_span.txt 2>&1; tail -70 $O/timeline_span.txt
