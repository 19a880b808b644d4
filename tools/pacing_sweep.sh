#!/bin/bash
# sweep of the short-run pacing knob (development aid)
for p in "" "4,1" "8,1" "8,2" "16,1" "16,2" "32,1"; do
  if [ -z "$p" ]; then arg=""; else arg="--pacing $p"; fi
  python bench.py --steps 1024 --warmup 64 --no-cpu-baseline $arg 2>/dev/null | python -c "
import json,sys; d=json.load(sys.stdin); print('pacing=%-6s us/step %.2f  fwd %.2f bwd %.2f' % ('$p', d['ms_per_step']*1e3, d['kernels']['fwd_fused_kernel(gather+rank)']['avg_us'], d['kernels']['bwd_fused_kernel(sgd apply+finish)']['avg_us']))"
done
