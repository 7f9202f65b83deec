#!/usr/bin/env python3
"""Condense a rocprofv3 --kernel-trace --stats run (CSV output) into profiles/<name>.md.

usage: tools/profile_summary.py <dir with *_kernel_stats.csv> <steps | auto> <out.md> [note]

`auto` (what tools/profile_round.sh passes): the number of profiled train steps is the number of launches of
`embed_gather_kernel` -- K1 runs exactly once per train step, whether the step is launched eagerly, captured or replayed
from a HIP graph -- so "calls/step" of that row is 1.0 by construction and every other row is normalised by the steps that
are really in the trace (round 2 divided 51 traced steps by 39: VERDICT r2)."""
import csv
import glob
import sys


def main():
    src, steps, out = sys.argv[1], sys.argv[2], sys.argv[3]
    note = sys.argv[4] if len(sys.argv) > 4 else ""
    f = sorted(glob.glob(src + "/**/*kernel_stats.csv", recursive=True))[0]
    rows = list(csv.DictReader(open(f)))
    gathers = sum(int(r["Calls"]) for r in rows if "embed_gather_kernel" in r["Name"])
    if steps == "auto":
        if gathers <= 0:
            raise SystemExit("profile_summary: no embed_gather_kernel launch in %s -- cannot count the steps" % f)
        steps = gathers
    else:
        steps = int(steps)
        if gathers and gathers != steps:
            raise SystemExit("profile_summary: %d steps given but the trace holds %d embed_gather_kernel launches" % (steps, gathers))
    total = sum(float(r["TotalDurationNs"]) for r in rows)
    with open(out, "w") as o:
        o.write("# rocprofv3 --kernel-trace --stats summary\n\n%s\n\n" % note)
        o.write("source: `%s`, %d train steps in the trace (= launches of `embed_gather_kernel`), GPU-busy %.3f ms/step "
                "(all steps of the trace: eager warm-up and event-bracketed steps included, so this is not the replayed step's "
                "time -- the bench line is)\n\n" % (f, steps, total / 1e6 / steps))
        o.write("| kernel | calls/step | ms/step | avg us | % |\n|---|---:|---:|---:|---:|\n")
        for r in rows[:44]:
            o.write("| `%s` | %.2f | %.4f | %.1f | %.1f |\n" % (
                r["Name"][:96].replace("|", "/"), int(r["Calls"]) / steps, float(r["TotalDurationNs"]) / 1e6 / steps,
                float(r["AverageNs"]) / 1e3, float(r["Percentage"])))
    print("wrote", out)


if __name__ == "__main__":
    main()
