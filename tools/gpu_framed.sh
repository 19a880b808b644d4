#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
R=$(pwd); O=$R/gpurun_out/framed; mkdir -p $O
python tools/framed_prof.py 2>&1 | grep "us/step"; BLOCK=16 python tools/framed_prof.py 2>&1 | grep "us/step"; BLOCK=2 python tools/framed_prof.py 2>&1 | grep "us/step"
EAGER=1 python tools/framed_prof.py 2>&1 | grep "us/step"
HOSTPROF=1 python tools/framed_prof.py 2>&1 | grep -v amdgpu.ids | head -40
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $R/tools/framed_prof.py > $O/stats.log 2>&1
python3 - <<PY
import csv, glob
for f in glob.glob("$O/stats/*/*_kernel_stats.csv"):
    for r in csv.DictReader(open(f)):
        print("%-70s calls %6s avg_us %8.2f total_ms %8.2f" % (r["Name"][:70], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6))
PY
cd $R
MASTER_ADDR=127.0.0.1 MASTER_PORT=29633 RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 HA_FORCE_SHARDED=1 python3 bench.py --steps 500 --warmup 50 --no-cpu-baseline 2>/dev/null | grep '^{' | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('bench sharded world 1: us/step %.2f' % (d['ms_per_step']*1e3), d['config']['exchange'])"
