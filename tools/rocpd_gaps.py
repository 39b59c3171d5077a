"""Device timeline of a rocprofv3 kernel trace (rocpd sqlite): union busy time, concurrency and the largest idle gaps of the last steps.
usage: python tools/rocpd_gaps.py results.db [n_last_ms]"""
import sqlite3
import sys

db = sys.argv[1]
last_ms = float(sys.argv[2]) if len(sys.argv) > 2 else 600.0
c = sqlite3.connect(db)
cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
namecol = "name" if "name" in cols else [x for x in cols if "name" in x][0]
rows = c.execute(f"select start, end, {namecol} from kernels order by start").fetchall()
t_end = max(r[1] for r in rows)
rows = [r for r in rows if r[0] >= t_end - last_ms * 1e6]
t0, t1 = rows[0][0], max(r[1] for r in rows)
busy = 0
cur_s, cur_e = rows[0][0], rows[0][1]
gaps = []
last_name = rows[0][2]
for s, e, n in rows[1:]:
    if s > cur_e:
        busy += cur_e - cur_s
        gaps.append((s - cur_e, last_name[:60], n[:60]))
        cur_s, cur_e = s, e
        last_name = n
    else:
        if e > cur_e:
            cur_e = e
            last_name = n
busy += cur_e - cur_s
span = t1 - t0
ksum = sum(e - s for s, e, _ in rows)
print(f"window {span/1e6:.2f} ms: union busy {busy/1e6:.2f} ms ({100*busy/span:.1f} %), sum of kernel durations {ksum/1e6:.2f} ms (avg concurrency {ksum/busy:.2f}), idle {100*(1-busy/span):.1f} %")
gaps.sort(reverse=True)
tot = sum(g[0] for g in gaps)
print(f"{len(gaps)} gaps, total {tot/1e6:.2f} ms; > 20 us: {sum(1 for g in gaps if g[0] > 20e3)} totalling {sum(g[0] for g in gaps if g[0] > 20e3)/1e6:.2f} ms")
for g in gaps[:25]:
    print(f"  {g[0]/1e3:8.1f} us   after {g[1]}  -> before {g[2]}")
