#!/bin/bash
# round 4, GPU session U: the apply launch's prologue (header in one batch of loads, the first item beside it): A/B + parity
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r4u; mkdir -p $O
timeout 2400 python -m pytest tests/test_gpu_qstep.py tests/test_gpu_tolerance.py -x -q -m gpu > $O/t_qstep.log 2>&1; echo "qstep rc $?" >> $O/rc.txt
REPS=3 bash tools/ab_variants.sh "hdr_batch+spec_item:" "round3_prologue:-DQV_HDR_BATCH=0" "hdr_batch_only:-DQV_SPEC_ITEM=0" > $O/ab.txt 2>&1
# leave the product build in place for anything that follows
FL="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-gpu-rdc -I include"
/opt/rocm/bin/hipcc $FL -c herald_amd/csrc/qstep.hip -o /tmp/qstep_final.o && /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -fno-gpu-rdc -o herald_amd/libherald_amd.so $(ls herald_amd/_build/*.o | grep -v qstep.o) /tmp/qstep_final.o
timeout 1500 python -m pytest tests/test_gpu_fullscale.py -x -q -m gpu -k "queue_step" > $O/t_full.log 2>&1; echo "fullscale rc $?" >> $O/rc.txt
cat $O/rc.txt $O/ab.txt
