#!/bin/bash
# round 4, GPU session H: stream-priority A/B, then the final profiles and the whole GPU test suite
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4h; mkdir -p $O
B="--no-cpu-baseline --no-cache-tier --no-laia --no-cold-tier --no-wide"
for i in 1 2; do
timeout 400 python bench.py $B 2>/dev/null | python tools/ab_line.py base long >> $O/prio.txt
timeout 400 python bench.py $B --steps 20 --warmup 5 2>/dev/null | python tools/ab_line.py base short >> $O/prio.txt
HA_QSIDE_PRIO=low HA_BENCH_MAIN_PRIO=high timeout 400 python bench.py $B 2>/dev/null | python tools/ab_line.py prio long >> $O/prio.txt
HA_QSIDE_PRIO=low HA_BENCH_MAIN_PRIO=high timeout 400 python bench.py $B --steps 20 --warmup 5 2>/dev/null | python tools/ab_line.py prio short >> $O/prio.txt
done
timeout 3000 bash tools/final_profiles_r4.sh > $O/final.log 2>&1
timeout 2400 python -m pytest tests -q -m gpu -x > $O/t_all.log 2>&1; echo "all gpu tests rc $?" >> $O/rc.txt
tail -3 $O/t_all.log
ls $O gpurun_out/final_r04/summary
