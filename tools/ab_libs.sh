#!/bin/bash
# same-box A/B of variant libraries ab/v/libherald_amd_<name>.so against the built one: the default bench command, the library
# file swapped in place between the runs (the box copy only)
B="--no-cpu-baseline --no-cache-tier --no-laia --no-cold-tier --no-wide"
cd $GRAFT_REPO_ROOT
cp herald_amd/libherald_amd.so /tmp/lib_head.so
for rep in 1 2; do
  for t in "$@"; do
    if [ "$t" = head ]; then cp /tmp/lib_head.so herald_amd/libherald_amd.so; else cp ab/v/libherald_amd_$t.so herald_amd/libherald_amd.so; fi
    python3 bench.py $B $BENCH_EXTRA 2>/dev/null | python3 tools/ab_line.py $t long
  done
done
cp /tmp/lib_head.so herald_amd/libherald_amd.so
(cd ab/r4 && python3 bench.py $B 2>/dev/null | python3 $GRAFT_REPO_ROOT/tools/ab_line.py r4_tree long)
