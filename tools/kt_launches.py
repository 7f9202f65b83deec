#!/usr/bin/env python3
"""Per-launch durations of the kernels whose name contains a pattern, from a rocprofv3 --kernel-trace CSV:
tools/kt_launches.py <dir> <pattern> [max launches]  ->  grid, block, duration (us) of every launch + the median per grid."""
import collections
import csv
import glob
import statistics
import sys

d, pat = sys.argv[1], sys.argv[2]
lim = int(sys.argv[3]) if len(sys.argv) > 3 else 12
f = sorted(glob.glob(d + "/**/*kernel_trace.csv", recursive=True))[0]
by = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    if pat in r["Kernel_Name"]:
        key = (r["Kernel_Name"][:60], r.get("Grid_Size_X", "?"), r.get("Grid_Size_Y", "?"), r.get("Workgroup_Size_X", "?"))
        by[key].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-3)
for k, v in sorted(by.items()):
    print("%-62s grid %s x %s wg %s: n=%d median %.1f us  min %.1f  max %.1f" % (k + (len(v), statistics.median(v), min(v), max(v))))
