#!/bin/bash
# same-box A/B of whole trees under ab/<name> (git archive of a commit, built in place) and HEAD: the default bench command
# (head_spanq = HEAD with the spanning launch's three-batch queue builder, HA_QSPAN_QUEUES=1)
B="--no-cpu-baseline --no-cache-tier --no-laia --no-cold-tier --no-wide"
cd $GRAFT_REPO_ROOT
for rep in 1 2; do
  for t in "$@"; do
    if [ "$t" = head ]; then d=.; e=0; elif [ "$t" = head_spanq ]; then d=.; e=1; else d=ab/$t; e=0; fi
    (cd $d && HA_QSPAN_QUEUES=$e python3 bench.py $B 2>/dev/null | python3 $GRAFT_REPO_ROOT/tools/ab_line.py $t long)
  done
done
