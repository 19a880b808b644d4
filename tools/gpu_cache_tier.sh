#!/bin/bash
# Cache tier (configs[1]: LRU, limit 0.1 x rows): parity tests, the lookup + update pair timed with each fusion on / off
# (HA_CACHE_FUSED bit 0 = two-launch update, bit 1 = eviction beside the lookup's row copies), and the per-kernel profile.
# Run through gpurun; results under gpurun_out/cache_tier/.
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/cache_tier; mkdir -p $O; rm -f $O/ab.txt
if [ "$1" != "notest" ]; then
timeout 900 python -m pytest tests/test_gpu_cache.py tests/test_gpu_cache_remote.py tests/test_gpu_hetu_ops.py -x -q > $O/pytest.log 2>&1; tail -4 $O/pytest.log
fi
B="--no-cpu-baseline --no-laia --no-cold-tier --no-wide --no-config-c --steps 256 --warmup 64"
for rep in 1 2; do for f in ${MODES:-3 2 1 0}; do
  HA_CACHE_FUSED=$f python bench.py $B 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('HA_CACHE_FUSED=$f cache_tier us/pair %.2f' % d['cache_tier']['us_per_step'])" | tee -a $O/ab.txt
done; done
for f in ${MODES:-3 0}; do
  HA_CACHE_FUSED=$f python bench.py $B --no-cache-prefill 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('HA_CACHE_FUSED=$f cache far from full (no prefill) us/pair %.2f' % d['cache_tier']['us_per_step'])" | tee -a $O/ab.txt
done
timeout 600 python tools/cache_phases.py 2>/dev/null > $O/cache_phases.txt; cat $O/cache_phases.txt
export TMPDIR=/tmp
rm -rf /tmp/ctprof; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ctprof -- python3 bench.py --no-cpu-baseline --no-laia --no-cold-tier --no-wide --no-config-c --steps 64 --warmup 32 > $O/bench_under_rocprof.json 2>$O/rocprof.err
python - <<'PY'
import csv, glob
f = glob.glob("/tmp/ctprof/**/*kernel_stats.csv", recursive=True)
rows = list(csv.DictReader(open(f[0])))
with open("gpurun_out/cache_tier/cache_tier_kernel_stats.csv", "w") as o:
    o.write("kernel,calls,avg_us,total_ms\n")
    for r in rows[:16]:
        o.write('"%s",%s,%.2f,%.3f\n' % (r["Name"][:70], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6))
print(open("gpurun_out/cache_tier/cache_tier_kernel_stats.csv").read())
PY
