#!/bin/bash
# round 4, GPU session H: the final profiles and the whole GPU test suite
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4h; mkdir -p $O
timeout 3000 bash tools/final_profiles_r4.sh > $O/final.log 2>&1
timeout 3000 python -m pytest tests -q -m gpu -x > $O/t_all.log 2>&1; echo "all gpu tests rc $?" >> $O/rc.txt
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; echo "smoke rc $?" >> $O/rc.txt
tail -3 $O/t_all.log; cat $O/rc.txt
ls gpurun_out/final_r04/summary
