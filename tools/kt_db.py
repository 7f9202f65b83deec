"""Per-kernel totals from a rocprofv3 --kernel-trace run kept as rocpd database:  python tools/kt_db.py <dir> [steps] [top]"""
import glob
import sqlite3
import sys

src, steps, top = sys.argv[1], float(sys.argv[2]) if len(sys.argv) > 2 else 1.0, int(sys.argv[3]) if len(sys.argv) > 3 else 25
db = sqlite3.connect(sorted(glob.glob(src + "/**/*.db", recursive=True))[0])
rows = list(db.execute("select name, count(*), sum(end-start)/1e3, min(end-start)/1e3 from kernels group by name order by 3 desc"))
tot = sum(r[2] for r in rows)
print("all kernels: %.3f ms per step (%d steps)" % (tot / steps / 1e3, steps))
for n, k, t, mn in rows[:top]:
    print("%-70s %7.1f launches/step %9.1f us/step  avg %8.1f us  min %8.1f" % (n[:70], k / steps, t / steps, t / k, mn))
