#!/bin/bash
# round 4, GPU session V: the world-1 sharded step by routing block size
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r4v; mkdir -p $O
export MASTER_ADDR=127.0.0.1 MASTER_PORT=29633 RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 HA_FORCE_SHARDED=1
for blk in 8 16 32; do
HA_SHARD_BLOCK=$blk timeout 900 python bench.py --steps 500 --warmup 80 --no-cpu-baseline 2>/dev/null | grep '^{' > $O/sharded_b$blk.json
HA_SHARD_BLOCK=$blk timeout 600 python tools/framed_hostprof.py 2>&1 | grep "us/step" | sed "s/^/block $blk: /" >> $O/host.txt
done
python - <<'PY'
import json
for b in (8, 16, 32):
    d = json.loads(open("gpurun_out/r4v/sharded_b%d.json" % b).readline())
    print("block", b, "config A %.2f us" % (d["ms_per_step"] * 1e3), "config_c %.2f us" % (d["config_c"]["ms_per_step"] * 1e3))
PY
cat $O/host.txt
