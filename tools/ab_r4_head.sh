#!/bin/bash
# same-box A/B: the tree of round 4's last commit (ab/r4, built in place) against HEAD, the default bench command, alternating
B="--no-cpu-baseline --no-cache-tier --no-laia --no-cold-tier --no-wide"
cd $GRAFT_REPO_ROOT
for rep in 1 2 3; do
  (cd ab/r4 && python3 bench.py $B 2>/dev/null | python3 $GRAFT_REPO_ROOT/tools/ab_line.py round4_tree long)
  python3 bench.py $B 2>/dev/null | python3 tools/ab_line.py head long
  (cd ab/r4 && python3 bench.py $B --steps 20 --warmup 5 2>/dev/null | python3 $GRAFT_REPO_ROOT/tools/ab_line.py round4_tree short)
  python3 bench.py $B --steps 20 --warmup 5 2>/dev/null | python3 tools/ab_line.py head short
done
