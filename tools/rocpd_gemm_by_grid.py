"""GEMM kernels of a rocprofv3 (rocpd sqlite) kernel trace grouped by (kernel, grid size) -- the grid identifies the problem shape.
usage: python tools/rocpd_gemm_by_grid.py results.db [steps]"""
import re
import sqlite3
import sys

db = sys.argv[1]
steps = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
c = sqlite3.connect(db)
cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
namecol = "name" if "name" in cols else [x for x in cols if "name" in x][0]
gcol = [x for x in cols if "grid" in x and "x" in x.lower()]
gcol = gcol[0] if gcol else None
wcol = [x for x in cols if "workgroup" in x and "x" in x.lower()]
wcol = wcol[0] if wcol else None
q = f"select {namecol}, {gcol or 0}, {wcol or 1}, count(*), sum(end-start) from kernels where {namecol} like '%gemm%' group by 1, 2, 3 order by 5 desc"
rows = c.execute(q).fetchall()
tot = sum(r[4] for r in rows)
print(f"GEMM kernel time {tot / 1e6 / steps:.2f} ms per step ({steps:g} steps); columns: kernel | workgroups | launches/step | ms/step | avg us")
fam = {}
for n, g, w, cnt, t in rows:
    short = re.sub(r"\(.*$", "", n).replace("void ", "")
    fam[short] = fam.get(short, 0) + t
for k, v in sorted(fam.items(), key=lambda x: -x[1]):
    print(f"  {k[:70]:70s} {v / 1e6 / steps:8.2f} ms/step")
print()
for n, g, w, cnt, t in rows[:60]:
    short = re.sub(r"\(.*$", "", n).replace("void ", "")
    print(f"{short[:62]:62s} {int(g) // max(int(w), 1):6d} {cnt / steps:7.1f} {t / 1e6 / steps:8.3f} {t / cnt / 1e3:8.1f}")
