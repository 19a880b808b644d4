#!/bin/bash
# round 4, GPU session T: the next launch's cold rows fetched by this launch's finished waves (experiment); block sizes
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r4t; mkdir -p $O
B="--no-cpu-baseline --no-cache-tier --no-laia --no-cold-tier --no-wide"
for i in 1 2 3; do
for pf in 0 1; do
HA_QPREFETCH=$pf timeout 400 python bench.py $B 2>/dev/null | python tools/ab_line.py "prefetch $pf" long >> $O/ab.txt
HA_QPREFETCH=$pf timeout 400 python bench.py $B --steps 20 --warmup 5 2>/dev/null | python tools/ab_line.py "prefetch $pf" short >> $O/ab.txt
done
done
for blk in 8 32; do
timeout 400 python bench.py $B --queue-block $blk 2>/dev/null | python tools/ab_line.py "block $blk" long >> $O/ab.txt
timeout 400 python bench.py $B --queue-block $blk --steps 20 --warmup 5 2>/dev/null | python tools/ab_line.py "block $blk" short >> $O/ab.txt
done
HA_QPREFETCH=1 timeout 900 python -m pytest tests/test_gpu_qstep.py -x -q -m gpu -k "criteo or small_tables and 512" > $O/t_pf.log 2>&1; echo "prefetch tests rc $?" >> $O/rc.txt
cat $O/ab.txt $O/rc.txt
