#!/usr/bin/env python3
"""Print a rocprofv3 *_kernel_stats.csv (found under the directory given) as: total ms, calls, average us, kernel."""
import csv, glob, os, sys
f = max(glob.glob(os.path.join(sys.argv[1], "**", "*kernel_stats.csv"), recursive=True), key=os.path.getmtime)
for r in csv.DictReader(open(f)):
    print("%9.3f ms %5s  %9.1f us  %s" % (int(r["TotalDurationNs"]) / 1e6, r["Calls"], float(r["AverageNs"]) / 1e3, r["Name"].split("(")[0][:60]))
