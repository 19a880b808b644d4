#!/bin/bash
# tools/ab_trees.sh for the driver's short run (--steps 20 --warmup 5)
B="--no-cpu-baseline --no-cache-tier --no-laia --no-cold-tier --no-wide --steps 20 --warmup 5"
cd $GRAFT_REPO_ROOT
for rep in 1 2 3; do
  for t in "$@"; do
    if [ "$t" = head ]; then d=.; else d=ab/$t; fi
    (cd $d && python3 bench.py $B 2>/dev/null | python3 $GRAFT_REPO_ROOT/tools/ab_line.py $t short)
  done
done
