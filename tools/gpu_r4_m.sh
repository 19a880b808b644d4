#!/bin/bash
# round 4, GPU session M: what the 20-step run pays that the long run does not; cache tier with / without the early sort
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r4m; mkdir -p $O
B="--no-cpu-baseline --no-cache-tier --no-laia --no-cold-tier --no-wide"
for i in 1 2; do
timeout 400 python bench.py $B --steps 20 --warmup 5 2>/dev/null | python tools/ab_line.py "w5 (driver)" short >> $O/short.txt
timeout 400 python bench.py $B --steps 20 --warmup 101 --pre-roll 96 2>/dev/null | python tools/ab_line.py "w101 pre-roll 96" short >> $O/short.txt
timeout 400 python bench.py $B --steps 20 --warmup 5 --clock-warm 64 2>/dev/null | python tools/ab_line.py "w5 clock-warm 64" short >> $O/short.txt
timeout 400 python bench.py $B --steps 20 --warmup 5 --clock-warm 0 2>/dev/null | python tools/ab_line.py "w5 clock-warm 0" short >> $O/short.txt
timeout 400 python bench.py $B --steps 20 --warmup 5 --no-gate 2>/dev/null | python tools/ab_line.py "w5 no gate" short >> $O/short.txt
HA_QSYNC=events timeout 400 python bench.py $B --steps 20 --warmup 5 2>/dev/null | python tools/ab_line.py "w5 events" short >> $O/short.txt
timeout 400 python bench.py $B --steps 16 --warmup 16 2>/dev/null | python tools/ab_line.py "w16 s16 (no boundary inside?)" short >> $O/short.txt
timeout 400 python bench.py $B --steps 100 --warmup 5 2>/dev/null | python tools/ab_line.py "w5 s100" short >> $O/short.txt
timeout 400 python bench.py $B 2>/dev/null | python tools/ab_line.py "default" long >> $O/short.txt
done
HA_CACHE_BENCH_AHEAD=0 timeout 900 python bench.py --no-cpu-baseline --no-laia --no-cold-tier --no-wide 2>/dev/null | grep '^{' > $O/bench_cache_noahead.json
HA_CACHE_BENCH_AHEAD=1 timeout 900 python bench.py --no-cpu-baseline --no-laia --no-cold-tier --no-wide 2>/dev/null | grep '^{' > $O/bench_cache_ahead.json
cat $O/short.txt
python - <<'PY'
import json
for f in ("noahead", "ahead"):
    d = json.loads(open("gpurun_out/r4m/bench_cache_%s.json" % f).readline())
    print(f, d.get("cache_tier", {}).get("us_per_step"))
PY
