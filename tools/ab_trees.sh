#!/bin/bash
# same-box A/B of whole trees under ab/<name> (git archive of a commit, built in place) and HEAD: the default bench command
B="--no-cpu-baseline --no-cache-tier --no-laia --no-cold-tier --no-wide"
cd $GRAFT_REPO_ROOT
for rep in 1 2; do
  for t in "$@"; do
    if [ "$t" = head ]; then d=.; else d=ab/$t; fi
    (cd $d && python3 bench.py $B 2>/dev/null | python3 $GRAFT_REPO_ROOT/tools/ab_line.py $t long)
  done
done
