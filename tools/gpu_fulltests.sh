#!/bin/bash
O=gpurun_out/fulltests; mkdir -p $O
timeout 5000 python -m pytest tests -m gpu -x -q > $O/t_all.log 2>&1; tail -5 $O/t_all.log | cut -c1-300
python - <<'PY'
import __graft_entry__ as g
g.smoke(); print("smoke ok")
PY
