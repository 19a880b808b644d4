#!/bin/bash
O=$GRAFT_REPO_ROOT/gpurun_out/r5x; mkdir -p $O
cd $GRAFT_REPO_ROOT
for i in 1 2 3; do python3 tools/laia_profile.py 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print({k:round(v,1) for k,v in d.items() if '_us' in k or 'us_per' in k})"; done
