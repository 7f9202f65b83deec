#!/usr/bin/env python3
"""Condense a rocprofv3 --kernel-trace --stats run (CSV output) into profiles/<name>.md.

usage: tools/profile_summary.py <dir with *_kernel_stats.csv> <steps profiled> <out.md> [note]"""
import csv
import glob
import sys


def main():
    src, steps, out = sys.argv[1], int(sys.argv[2]), sys.argv[3]
    note = sys.argv[4] if len(sys.argv) > 4 else ""
    f = sorted(glob.glob(src + "/**/*kernel_stats.csv", recursive=True))[0]
    rows = list(csv.DictReader(open(f)))
    total = sum(float(r["TotalDurationNs"]) for r in rows)
    with open(out, "w") as o:
        o.write("# rocprofv3 --kernel-trace --stats summary\n\n%s\n\n" % note)
        o.write("source: `%s`, %d profiled steps, GPU-busy %.3f ms/step\n\n" % (f, steps, total / 1e6 / steps))
        o.write("| kernel | calls/step | ms/step | avg us | % |\n|---|---:|---:|---:|---:|\n")
        for r in rows[:40]:
            o.write("| `%s` | %.1f | %.4f | %.1f | %.1f |\n" % (
                r["Name"][:96].replace("|", "/"), int(r["Calls"]) / steps, float(r["TotalDurationNs"]) / 1e6 / steps,
                float(r["AverageNs"]) / 1e3, float(r["Percentage"])))
    print("wrote", out)


if __name__ == "__main__":
    main()
