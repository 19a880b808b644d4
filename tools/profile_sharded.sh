#!/bin/bash
# rocprofv3 evidence of the sharded step at world size 1 (run through gpurun): kernel stats + PMC traffic
TAG=${1:-r03}
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof_sharded_$TAG
mkdir -p $OUT
export MASTER_ADDR=127.0.0.1 MASTER_PORT=29633 RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 HA_FORCE_SHARDED=1
cd /tmp && export TMPDIR=/tmp
A="--steps 400 --warmup 60 --no-cpu-baseline --no-config-c"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $R/bench.py $A > $OUT/stats.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $R/bench.py --steps 64 --warmup 60 --no-cpu-baseline --no-config-c > $OUT/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 $R/bench.py --steps 64 --warmup 60 --no-cpu-baseline --no-config-c > $OUT/pmc_write.log 2>&1
python3 $R/tools/pmc_summary.py $OUT $OUT/summary > $OUT/summary.log 2>&1
grep '^{' $OUT/stats.log | tail -1 > $OUT/summary/bench_sharded_world1_under_rocprof.json
cd $R
python3 bench.py --steps 500 --warmup 60 --no-cpu-baseline 2>/dev/null | grep '^{' > $OUT/summary/bench_sharded_world1.json
HA_SHARD_GRAPHS=1 python3 bench.py --steps 500 --warmup 60 --no-cpu-baseline --no-config-c 2>/dev/null | grep '^{' > $OUT/summary/bench_sharded_world1_graphs.json
HA_SHARD_SIZED=1 python3 bench.py --steps 500 --warmup 60 --no-cpu-baseline --no-config-c 2>/dev/null | grep '^{' > $OUT/summary/bench_sharded_world1_sized.json
for f in bench_sharded_world1 bench_sharded_world1_graphs bench_sharded_world1_sized; do python3 - <<PY
import json
try:
    d=json.loads(open("$OUT/summary/$f.json").read())
    print("$f: us/step %.2f |" % (d["ms_per_step"]*1e3), d["config"].get("exchange"))
    k=(d["roofline"].get("kernels") or {})
    for n,v in k.items(): print("    %-58s %6.2f us  %7.1f GB/s  frac %.3f traffic %s" % (n, v["us"], v["GBps"], v["frac_of_hbm_peak"], v["traffic"]))
    c=d.get("config_c")
    if c:
        print("  config_c:", c.get("error") or ("us/step %.2f | %s" % (c["ms_per_step"]*1e3, c["config"]["exchange"])))
        for n,v in ((c.get("roofline") or {}).get("kernels") or {}).items(): print("    %-58s %6.2f us  %7.1f GB/s  frac %.3f" % (n, v["us"], v["GBps"], v["frac_of_hbm_peak"]))
except Exception as e: print("$f: no result", e)
PY
done
cat $OUT/summary/bench_kernel_stats.csv | cut -c1-150
