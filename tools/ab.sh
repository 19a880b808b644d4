#!/bin/bash
# A/B of library builds kept under ab/ (development aid): rocprofv3 kernel averages + bench per variant
cd /tmp && export TMPDIR=/tmp
for v in "$@"; do
  cp $GRAFT_REPO_ROOT/ab/lib$v.so $GRAFT_REPO_ROOT/herald_amd/libherald_amd.so
  echo "== variant $v"
  if [ -n "$AB_KSTATS" ]; then
    rm -rf /tmp/ab_$v
    rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ab_$v -- python3 $GRAFT_REPO_ROOT/tools/kbench.py --case criteo > /tmp/ab_$v.log 2>&1
    python3 $GRAFT_REPO_ROOT/tools/kstats.py /tmp/ab_$v | grep -E "$AB_KSTATS"
  fi
  (cd $GRAFT_REPO_ROOT && python3 bench.py --no-cpu-baseline --no-cache-tier 2>/dev/null | tail -1 | python3 tools/ab_line.py)
done
