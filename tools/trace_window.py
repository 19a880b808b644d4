#!/usr/bin/env python3
"""A window of a rocprofv3 --kernel-trace (csv) as one timeline over all streams: start / end / duration / queue / kernel, and
the idle time of the busiest queue (development aid).
usage: trace_window.py <trace dir> [fraction of the trace where the window starts = 0.7] [dispatches = 60]"""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
frac = float(sys.argv[2]) if len(sys.argv) > 2 else 0.7
count = int(sys.argv[3]) if len(sys.argv) > 3 else 60
rows = []
for r in csv.DictReader(open(f)):
    name = r["Kernel_Name"].split("(")[0].replace("ha::", "").replace("void ", "")
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), name[:44], r.get("Queue_Id", "?")))
rows.sort()
i0 = int(len(rows) * frac)
t0 = rows[i0][0]
for r in rows[i0:i0 + count]:
    print("  %9.1f %9.1f  %7.1f us  queue %-3s %s" % ((r[0] - t0) / 1e3, (r[1] - t0) / 1e3, (r[1] - r[0]) / 1e3, r[3], r[2]))
sub = rows[i0:]
byq = {}
for r in sub:
    byq.setdefault(r[3], []).append(r)
for q, v in sorted(byq.items(), key=lambda kv: -len(kv[1])):
    busy = sum(e - s for s, e, _, _ in v)
    span = v[-1][1] - v[0][0]
    print("queue %s: %d dispatches, busy %.1f %% of its span (%.1f us), mean gap %.2f us" % (
        q, len(v), 100.0 * busy / max(span, 1), span / 1e3, (span - busy) / 1e3 / max(len(v) - 1, 1)))
