#!/usr/bin/env python3
"""Where the laia scheduler's stream is idle: the kernel trace of tools/laia_profile.py (rocprofv3 --kernel-trace) as one
timeline -- per global batch the sum of kernel durations and the period, and the raw timeline of two batches.
usage: laia_gaps.py <dir with *_kernel_trace.csv>"""
import csv, glob, sys
import numpy as np
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("ha::", ""))
        for r in csv.DictReader(open(f))]
rows.sort()
starts = [i for i, r in enumerate(rows) if "laia_probe" in r[2]]      # a batch starts at its probe kernel
starts = starts[len(starts) // 2:]      # steady state
period = [rows[b][0] - rows[a][0] for a, b in zip(starts[:-1], starts[1:])]
busy = [sum(e - s for s, e, _ in rows[a:b]) for a, b in zip(starts[:-1], starts[1:])]
print("batches %d: period mean %.1f us, kernels %.1f us, stream idle %.1f us per batch"
      % (len(period), np.mean(period) / 1e3, np.mean(busy) / 1e3, (np.mean(period) - np.mean(busy)) / 1e3))
i0 = starts[len(starts) // 2]
t0 = rows[i0][0]
print("two batches, start / end in us (the tracer's stamps are good to a microsecond or two):")
for r in rows[i0:starts[len(starts) // 2 + 2]]:
    print("  %8.1f %8.1f  %s" % ((r[0] - t0) / 1e3, (r[1] - t0) / 1e3, r[2]))
