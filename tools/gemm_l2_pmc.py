"""Join a rocprofv3 --pmc pass over `tools/gemm_l2_ab.py pmc order.json` with its launch order.
usage: python tools/gemm_l2_pmc.py order.json out.json COUNTER=results.db [COUNTER=results.db ...]
Every (shape, config) was launched exactly once after one warm-up launch per shape; dispatches of the 8-phase kernel are taken in
dispatch order, the first of each shape (the warm-up) is dropped. FETCH_SIZE (KB) is doubled (gfx950 tallies 128-B requests at 64 B,
MI355X_MICROARCH.md); WRITE_SIZE (KB) as reported."""
import json
import sqlite3
import sys


def per_dispatch(db, name):
    c = sqlite3.connect(db)
    rows = c.execute("select dispatch_id, kernel_name, counter_name, value from counters_collection").fetchall()
    per = {}
    for did, kn, cn, val in rows:
        if cn == name and "gemm_8ph" in kn:
            per[did] = per.get(did, 0.0) + val
    return [per[k] for k in sorted(per)]


def main():
    order = json.load(open(sys.argv[1]))
    out = [dict(o) for o in order]
    for spec in sys.argv[3:]:
        ctr, db = spec.split("=", 1)
        vals = per_dispatch(db, ctr)
        shapes = []
        for o in order:
            if o["shape"] not in shapes:
                shapes.append(o["shape"])
        n_expected = len(order) + len(shapes)
        if len(vals) != n_expected:
            print(f"{ctr}: {len(vals)} dispatches, expected {n_expected}", file=sys.stderr)
            continue
        it = iter(vals)
        last = None
        for o in out:
            if o["shape"] != last:
                next(it)            # warm-up launch of this shape
                last = o["shape"]
            v = next(it)
            o[ctr] = v
    for o in out:
        M, N, K = o["MNK"]
        alg = 2.0 * (M * K + N * K + M * N * (2 if "deriv" in o["shape"] and "dgrad" in o["shape"] else 1))
        if "FETCH_SIZE" in o:
            o["fetch_bytes"] = 2.0 * o["FETCH_SIZE"] * 1024
        if "WRITE_SIZE" in o:
            o["write_bytes"] = o["WRITE_SIZE"] * 1024
        if "fetch_bytes" in o and "write_bytes" in o:
            o["traffic_over_algorithmic"] = round((o["fetch_bytes"] + o["write_bytes"]) / alg, 3)
        o["algorithmic_bytes"] = alg
        if "TCC_HIT_sum" in o and "TCC_MISS_sum" in o:
            o["l2_hit_rate"] = round(o["TCC_HIT_sum"] / max(o["TCC_HIT_sum"] + o["TCC_MISS_sum"], 1.0), 4)
        print({k: (round(v / 1e6, 1) if k.endswith("_bytes") else v) for k, v in o.items() if k not in ("MNK", "FETCH_SIZE", "WRITE_SIZE")})
    json.dump(out, open(sys.argv[2], "w"), indent=1)


if __name__ == "__main__":
    main()
