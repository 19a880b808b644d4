#!/bin/bash
# Hardware-counter passes over the default bench command's timed kernel (ha::qapply_kernel): what do its waves wait on?
# Every group is its own rocprofv3 run (--pmc only, as the guide prescribes); a group whose counter names this ROCm
# does not know fails alone.  Run through gpurun from the repo root; condensed by tools/pmc_wait_summary.py.
TAG=${1:-r04}
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmcw_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > $OUT/counters_available.txt 2>&1
ARGS="--steps 64 --warmup 32 --no-cpu-baseline --no-kernel-pass --no-cache-tier --no-cold-tier --no-laia --no-wide"
i=0
while read -r group; do
  [ -z "$group" ] && continue
  i=$((i+1))
  timeout 600 rocprofv3 --pmc $group --output-format csv -d $OUT/g$i -- python3 $GRAFT_REPO_ROOT/bench.py $ARGS > $OUT/g$i.log 2>&1
  echo "group $i: $group -> rc $?" >> $OUT/groups.txt
done <<'GROUPS'
SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR
SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VALU SQ_INSTS_SMEM SQ_INSTS_VALU SQ_INST_LEVEL_VMEM SQ_LEVEL_WAVES SQ_BUSY_CU_CYCLES
TCP_PENDING_STALL_CYCLES TCP_TCP_TA_DATA_STALL_CYCLES TCP_TA_TCP_STATE_READ TCP_GATE_EN1
TCP_TCC_READ_REQ TCP_TCC_WRITE_REQ TCP_TCC_READ_REQ_LATENCY TCP_TCC_WRITE_REQ_LATENCY
TCP_TOTAL_CACHE_ACCESSES TCP_TCC_NC_READ_REQ TCP_TCR_TCP_STALL_CYCLES TCP_READ_TAGCONFLICT_STALL_CYCLES
TCC_HIT TCC_MISS TCC_REQ TCC_READ
TCC_EA0_RDREQ TCC_EA0_WRREQ TCC_EA0_RDREQ_32B TCC_EA0_WRREQ_64B
TCC_EA0_WRREQ_STALL TCC_EA0_RD_UNCACHED_32B TCC_TAG_STALL TCC_BUSY
TCC_EA0_RDREQ_LEVEL TCC_EA0_WRREQ_LEVEL TCC_NORMAL_WRITEBACK TCC_ALL_TC_OP_WB_WRITEBACK
TA_BUSY TA_TA_BUSY TA_ADDR_STALLED_BY_TC_CYCLES TA_DATA_STALLED_BY_TC_CYCLES
TD_TD_BUSY TD_TC_STALL TCC_TOO_MANY_EA_WRREQS_STALL TCC_SRC_FIFO_FULL
GRBM_GUI_ACTIVE GRBM_COUNT
GROUPS
ls $OUT
python3 $GRAFT_REPO_ROOT/tools/pmc_wait_summary.py $OUT $OUT/summary.json
