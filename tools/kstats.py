#!/usr/bin/env python3
"""Print per-kernel averages of a rocprofv3 --stats output directory (development aid)."""
import csv, glob, sys
for f in glob.glob(sys.argv[1] + "/*/*_kernel_stats.csv"):
    for r in csv.DictReader(open(f)):
        if "ha::" in r["Name"]:
            print("%-46s calls %5s avg %8.2f us  min %7.2f max %7.2f" % (r["Name"].replace("void ", "")[:46], r["Calls"],
                  float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3))
