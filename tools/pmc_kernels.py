"""Join the rocprofv3 passes of tools/kernels_one.py into one per-kernel table: kernel-trace durations + SQ counters + FETCH / WRITE sizes.
usage: python tools/pmc_kernels.py out.md trace.db sq.db fetch.db write.db
Derived columns: wait % = SQ_WAIT_ANY / SQ_WAVE_CYCLES (waves parked at s_waitcnt / barriers), issue-stall % = SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES,
VALU % = SQ_ACTIVE_INST_VALU / SQ_WAVE_CYCLES, LDS % likewise (all quad-cycle units, MI355X_MICROARCH.md); MFMA pipe busy % =
SQ_VALU_MFMA_BUSY_CYCLES (cycles) / (kernel duration x ASSUMED_MHZ x 1024 SIMDs) -- the clock is not measured in these passes, 2100 MHz is
assumed; HBM-side bytes = 2 x FETCH_SIZE KB + WRITE_SIZE KB (gfx950 tallies 128-B fetch requests at 64 B; Infinity-Cache hits included).
The raw counters follow the table."""
ASSUMED_MHZ = 2100.0
import sqlite3
import sys


def counters(db):
    c = sqlite3.connect(db)
    out = {}
    for did, kn, cn, val in c.execute("select dispatch_id, kernel_name, counter_name, value from counters_collection"):
        out.setdefault((did, kn), {}).setdefault(cn, 0.0)
        out[(did, kn)][cn] += val
    return [(k[1], v) for k, v in sorted(out.items())]


def durations(db):
    c = sqlite3.connect(db)
    cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
    namecol = "name" if "name" in cols else [x for x in cols if "name" in x][0]
    return [(n, (e - s) / 1e3) for n, s, e in c.execute(f"select {namecol}, start, end from kernels order by start")]


def keep(name):
    return not any(x in name for x in ("at::native", "rocclr", "elementwise", "Memset", "fill"))


def main():
    out, trace, sq, fetch, write = sys.argv[1:6]
    dur = [(n, d) for n, d in durations(trace) if keep(n)]
    tabs = [[(n, v) for n, v in counters(db) if keep(n)] for db in (sq, fetch, write)]
    # every pass ran the same launch sequence: take the LAST len(sequence) kernels of each (warm-up launches come first)
    nseq = min(len(dur), *[len(t) for t in tabs])
    lines = ["# per-kernel counters of the non-GEMM hot kernels (tools/kernels_one.py under rocprofv3; one launch each at the bench shape)", "",
             "| kernel | us | HBM-side GB/s | fetch MB | write MB | wait % | issue-stall % | VALU % | MFMA pipe busy % (2.1 GHz assumed) | LDS inst % |", "|---|---|---|---|---|---|---|---|---|---|"]
    half = nseq // 2 if False else nseq
    seq = dur[-half:]
    tsq, tf, tw = (t[-half:] for t in tabs)
    for i, (n, d) in enumerate(seq):
        s, f, w = tsq[i][1], tf[i][1], tw[i][1]
        assert tsq[i][0] == n and tf[i][0] == n and tw[i][0] == n, (n, tsq[i][0], tf[i][0], tw[i][0])
        wc = max(s.get("SQ_WAVE_CYCLES", 0.0), 1.0)
        fb, wb = 2.0 * f.get("FETCH_SIZE", 0.0) * 1024, w.get("WRITE_SIZE", 0.0) * 1024
        mf = s.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (d * ASSUMED_MHZ * 1024.0)
        short = n.replace("void ", "")[:70]
        lines.append(f"| `{short}` | {d:.1f} | {(fb + wb) / d / 1e3:.0f} | {fb / 1e6:.1f} | {wb / 1e6:.1f} | {100 * s.get('SQ_WAIT_ANY', 0) / wc:.0f} | "
                     f"{100 * s.get('SQ_WAIT_INST_ANY', 0) / wc:.0f} | {100 * s.get('SQ_ACTIVE_INST_VALU', 0) / wc:.0f} | {100 * mf:.0f} | {100 * s.get('SQ_ACTIVE_INST_LDS', 0) / wc:.0f} |")
    lines += ["", "raw SQ counters (sums over the chip), same order:", ""]
    for i, (n, d) in enumerate(seq):
        lines.append(f"{i:2d} {n.replace('void ', '')[:60]:60s} " + " ".join(f"{k}={v:.4g}" for k, v in sorted(tsq[i][1].items())))
    txt = "\n".join(lines) + "\n"
    open(out, "w").write(txt)
    print(txt)


if __name__ == "__main__":
    main()
