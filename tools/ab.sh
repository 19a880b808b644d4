#!/bin/bash
# A/B of library builds kept under ab/ (development aid): rocprofv3 kernel averages + bench per variant
cd /tmp && export TMPDIR=/tmp
for v in "$@"; do
  cp $GRAFT_REPO_ROOT/ab/lib$v.so $GRAFT_REPO_ROOT/herald_amd/libherald_amd.so
  rm -rf /tmp/ab_$v
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ab_$v -- python3 $GRAFT_REPO_ROOT/tools/kbench.py --case criteo > /tmp/ab_$v.log 2>&1
  echo "== variant $v"; python3 $GRAFT_REPO_ROOT/tools/kstats.py /tmp/ab_$v | grep -E "rank|gather"
  (cd $GRAFT_REPO_ROOT && python3 bench.py --no-cpu-baseline --no-cache-tier 2>&1 | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(' bench us/step %.2f fwd %.2f bwd %.2f' % (d['ms_per_step']*1e3, d['kernels']['fwd_fused_kernel(gather+rank)']['avg_us'], d['kernels']['bwd_fused_kernel(sgd apply+finish)']['avg_us']))")
done
