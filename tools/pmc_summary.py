#!/usr/bin/env python3
"""Summarise a rocprofv3 --pmc run (CSV) per kernel: usage tools/pmc_summary.py <dir> [name filter]"""
import collections, csv, glob, sys
d = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else "cin_"
cc = sorted(glob.glob(d + "/**/*counter_collection.csv", recursive=True))[0]
kt = {}
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        kt[r["Dispatch_Id"]] = float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(cc)):
    n = r["Kernel_Name"]
    if flt not in n:
        continue
    key = n[:44] if n.startswith("_Z") else n.split("(")[0][-44:]      # mangled names (templates): the head tells the instance
    agg[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
    agg[key]["_dur_ns"].append(kt.get(r["Dispatch_Id"], 0.0))
for k, v in agg.items():
    print(k)
    for c, vals in sorted(v.items()):
        print("   %-30s %.5g" % (c, sum(vals) / len(vals)))
    if "GRBM_GUI_ACTIVE" in v and "SQ_VALU_MFMA_BUSY_CYCLES" in v:
        gui = sum(v["GRBM_GUI_ACTIVE"]) / len(v["GRBM_GUI_ACTIVE"])
        mf = sum(v["SQ_VALU_MFMA_BUSY_CYCLES"]) / len(v["SQ_VALU_MFMA_BUSY_CYCLES"])
        dur = sum(v["_dur_ns"]) / len(v["_dur_ns"])
        print("   -> clock %.3f GHz, MFMA pipe utilisation %.3f" % (gui / 8 / dur, mf / (gui / 8 * 1024)))
