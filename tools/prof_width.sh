#!/bin/bash
cd /tmp && export TMPDIR=/tmp
for w in 512 520 576 640; do
  rm -rf /tmp/pw_$w
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pw_$w -- python3 $GRAFT_REPO_ROOT/tools/kbench.py --rows 200000 --width $w --case hot3 > /tmp/pw_$w.log 2>&1
  echo "== width $w"; cat /tmp/pw_$w/*/*_kernel_stats.csv | grep "apply_kernel<0" | awk -F'","' '{print $2, $4}'
done
