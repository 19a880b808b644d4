#!/bin/bash
# rocprofv3 kernel stats of the unfused kernels on criteo-shaped batches (development aid)
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pa; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pa -- python3 $GRAFT_REPO_ROOT/tools/kbench.py --rows 2000000 --case criteo > /tmp/pa.log 2>&1
python3 $GRAFT_REPO_ROOT/tools/kstats.py /tmp/pa
