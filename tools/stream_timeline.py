#!/usr/bin/env python3
"""Both streams of a work-queue run as one timeline (rocprofv3 --kernel-trace csv): every dispatch with start / end, and per
apply launch its duration and what ran beside it.
usage: stream_timeline.py <trace dir> [first apply launch to print] [how many]"""
import csv, glob, sys
import numpy as np
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
first, count = int(sys.argv[2]) if len(sys.argv) > 2 else 40, int(sys.argv[3]) if len(sys.argv) > 3 else 12
rows = []
for r in csv.DictReader(open(f)):
    name = r["Kernel_Name"].split("(")[0].replace("ha::", "").replace("void ", "")
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), name[:28], r.get("Stream_Id", "?")))
rows.sort()
app = [i for i, r in enumerate(rows) if r[2].startswith("qapply")]
d = np.array([rows[i][1] - rows[i][0] for i in app]) / 1e3
gap = np.array([rows[b][0] - rows[a][1] for a, b in zip(app[:-1], app[1:])]) / 1e3
half = len(app) // 2
print("apply launches %d: duration mean %.1f p10 %.1f p50 %.1f p90 %.1f max %.1f us; gap to the next mean %.2f p90 %.2f max %.1f us (second half)"
      % (len(app), d[half:].mean(), *np.percentile(d[half:], [10, 50, 90]), d[half:].max(), gap[half:].mean(), np.percentile(gap[half:], 90), gap[half:].max()))
i0, i1 = app[first], app[min(first + count, len(app) - 1)]
t0 = rows[i0][0]
for r in rows[i0:i1 + 1]:
    print("  %9.1f %9.1f  %7.1f us  stream %-3s %s" % ((r[0] - t0) / 1e3, (r[1] - t0) / 1e3, (r[1] - r[0]) / 1e3, r[3], r[2]))
