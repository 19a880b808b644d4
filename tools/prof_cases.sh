#!/bin/bash
# rocprofv3 kernel stats of tools/kbench.py per id distribution (development aid)
cd /tmp && export TMPDIR=/tmp
for c in distinct criteo hot3 mid8; do
  rm -rf /tmp/pc_$c
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pc_$c -- python3 $GRAFT_REPO_ROOT/tools/kbench.py --rows 2000000 --case $c > /tmp/pc_$c.log 2>&1
  echo "== $c"; grep -v "^E2\|^W2" /tmp/pc_$c.log | tail -2
  cat /tmp/pc_$c/*/*_kernel_stats.csv | grep -v "at::native" | awk -F'","' '{print $1, $2, $4, $6, $7}' | cut -c1-160
done
