#!/usr/bin/env python3
"""bench.py's laia_scheduler leg alone (configs[3]'s shape: 4 workers x 1024 samples x 26 tables), for rocprofv3."""
import argparse, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
a = argparse.Namespace(fields=26, rows=33762577)
print(json.dumps(bench.laia_scheduler(a)))
