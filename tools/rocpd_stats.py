"""Summarise a rocprofv3 (rocpd sqlite) kernel trace: per-kernel calls / total / average duration.
usage: python tools/rocpd_stats.py results.db [out.md] [top_n]"""
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r"\(.*$", "", name)
    name = name.replace("void ", "")
    return name[:110]


def main():
    db = sys.argv[1]
    out = sys.argv[2] if len(sys.argv) > 2 else None
    top = int(sys.argv[3]) if len(sys.argv) > 3 else 40
    c = sqlite3.connect(db)
    cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
    namecol = "name" if "name" in cols else [x for x in cols if "name" in x][0]
    rows = c.execute(f"select {namecol}, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) from kernels group by {namecol} order by 3 desc").fetchall()
    total = sum(r[2] for r in rows)
    span = c.execute("select min(start), max(end) from kernels").fetchone()
    lines = [f"# rocprofv3 --kernel-trace summary ({db.split('/')[-1]})", "",
             f"total kernel time {total/1e6:.2f} ms over {sum(r[1] for r in rows)} dispatches; first->last span {(span[1]-span[0])/1e6:.2f} ms", "",
             "| kernel | calls | total ms | % | avg us | min us | max us |", "|---|---|---|---|---|---|---|"]
    for n, cnt, tot, avg, mn, mx in rows[:top]:
        lines.append(f"| `{short(n)}` | {cnt} | {tot/1e6:.2f} | {100*tot/total:.1f} | {avg/1e3:.1f} | {mn/1e3:.1f} | {mx/1e3:.1f} |")
    txt = "\n".join(lines) + "\n"
    if out:
        open(out, "w").write(txt)
    print(txt)


if __name__ == "__main__":
    main()
