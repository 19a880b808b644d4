#!/bin/bash
# round 4, GPU session J: side-stream priority on the wide shapes; world-1 sharded step after the host-path trim
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r4j; mkdir -p $O
for sh in "4096 128" "1024 512"; do set -- $sh
  for pr in none high low; do
    HA_QSIDE_PRIO=$pr BATCH=$1 WIDTH=$2 timeout 600 python tools/shape_bench.py 2>/dev/null | head -1 | sed "s/^/side prio $pr: /" >> $O/prio_shapes.txt
  done
done
B="--no-cpu-baseline --no-cache-tier --no-laia --no-cold-tier --no-wide"
HA_QSIDE_PRIO=high timeout 400 python bench.py $B 2>/dev/null | python tools/ab_line.py sidehigh long >> $O/prio.txt
HA_QSIDE_PRIO=high timeout 400 python bench.py $B --steps 20 --warmup 5 2>/dev/null | python tools/ab_line.py sidehigh short >> $O/prio.txt
timeout 400 python bench.py $B 2>/dev/null | python tools/ab_line.py base long >> $O/prio.txt
timeout 400 python bench.py $B --steps 20 --warmup 5 2>/dev/null | python tools/ab_line.py base short >> $O/prio.txt
HA_FORCE_SHARDED=1 timeout 900 python bench.py --no-cpu-baseline > $O/sharded_world1.json 2> $O/sharded_world1.err
timeout 600 python tools/framed_hostprof.py > $O/framed_hostprof.txt 2>&1
timeout 900 python -m pytest tests/test_gpu_framed.py -x -q -m gpu > $O/t_framed.log 2>&1; echo "framed rc $?" >> $O/rc.txt
cat $O/prio_shapes.txt $O/prio.txt $O/rc.txt; head -3 $O/framed_hostprof.txt; python - <<'PY'
import json
for l in open("gpurun_out/r4j/sharded_world1.json"):
    if l.startswith("{"):
        d = json.loads(l); print(d["ms_per_step"], d.get("config_c", {}).get("ms_per_step"))
PY
