"""HBM-side traffic of the dominant GEMM kernels -> profiles-style JSON (bench.py's roofline.traffic reads it).
Runs rocprofv3 --kernel-trace --pmc FETCH_SIZE and --pmc WRITE_SIZE as SEPARATE passes (MI355X_MICROARCH.md: FETCH_SIZE takes 3 of
the 4 TCC slots) on tools/gemm_one.py for each shape and sums the counters per dispatch of the GEMM kernel.
FETCH_SIZE (KB, tallied at 64 B per 128-B request on gfx950) is doubled; WRITE_SIZE (KB) is taken as reported.
usage (GPU box, from the repo root): python tools/pmc_gemm_traffic.py gpurun_out/r04_pmc_gemm_traffic.json"""
import glob
import json
import os
import sqlite3
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FAMILY = {1: "gemm_glds_kernel(128x128 LDS-DMA)", 3: "gemm_8ph_kernel(256x256 8-phase)", 4: "gemm_8ph2_kernel(256x128 8-phase, 2 workgroups per CU)"}
LAYOUT = {"NN": "forward x.W^T", "NT": "dgrad dY.W", "TT": "wgrad dY^T.X"}
SHAPES = [   # label, M N K ta tb -- the kernel family (hence bench.py's name of it) is asked from the library's policy at run time
    ("M=3072 N=768 K=100864 (ViT fc1 wgrad, 7 K-slices)", 3072, 768, 100864, 1, 1),
    ("M=100864 N=3072 K=768 (ViT fc1 forward)", 100864, 3072, 768, 0, 0),
    ("M=100864 N=768 K=3072 (ViT fc2 forward)", 100864, 768, 3072, 0, 0),
    ("M=100864 N=3072 K=768 (ViT fc2 dgrad)", 100864, 3072, 768, 0, 1),
    ("M=100864 N=768 K=3072 (ViT fc1 dgrad)", 100864, 768, 3072, 0, 1),
    ("M=16512 N=768 K=3072 (AST fc1 dgrad)", 16512, 768, 3072, 0, 1),
]


def bench_name(M, N, Kd, ta, tb):
    sys.path.insert(0, ROOT)
    from valor_amd import lib
    fam = lib.load().valor_gemm_kernel_for(0, ta, tb, M, N, Kd, 0)
    return f"{FAMILY.get(fam, 'family %d' % fam)} {'NT'[ta]}{'NT'[tb]} ({LAYOUT['NT'[ta] + 'NT'[tb]]})"


def counter(db, name):
    c = sqlite3.connect(db)
    rows = c.execute("select dispatch_id, kernel_name, counter_name, value from counters_collection").fetchall()
    per = {}
    for did, kn, cn, val in rows:
        if cn == name and "gemm_" in kn and "splitk_reduce" not in kn:
            per[did] = per.get(did, 0.0) + val
    vals = sorted(per.values())
    return vals[len(vals) // 2] if vals else None          # median over the repetitions


def main():
    out_path = sys.argv[1]
    work = "/tmp/pmc_gemm"
    res = {}
    for label, M, N, Kd, ta, tb in SHAPES:
        name = bench_name(M, N, Kd, ta, tb)
        if name in res:
            name = name + " | " + label
        got = {}
        for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
            d = f"{work}/{M}_{N}_{Kd}_{ta}{tb}_{ctr}"
            r = subprocess.run(["rocprofv3", "--kernel-trace", "--pmc", ctr, "-d", d, "-o", "p", "--", sys.executable, os.path.join(ROOT, "tools", "gemm_one.py"),
                                str(M), str(N), str(Kd), str(ta), str(tb), "3"], cwd=ROOT, env=dict(os.environ, TMPDIR="/tmp"),
                               capture_output=True, text=True, timeout=300)
            dbs = glob.glob(d + "/**/*.db", recursive=True)
            got[ctr] = counter(dbs[0], ctr) if dbs else None
            if got[ctr] is None:
                print(f"[{label} {ctr}] rc={r.returncode} dbs={dbs}\n" + (r.stdout + r.stderr)[-1500:], flush=True)
        if got["FETCH_SIZE"] is None or got["WRITE_SIZE"] is None:
            print("no counters for", label, got, flush=True)
            continue
        fetch, write = 2.0 * got["FETCH_SIZE"] * 1024, got["WRITE_SIZE"] * 1024
        alg = 2.0 * (M * Kd + N * Kd + M * N)
        res[name] = {"shape": label, "fetch_bytes": fetch, "write_bytes": write, "traffic_bytes": fetch + write, "algorithmic_bytes": alg}
        print(f"{label}: fetch {fetch / 1e6:.1f} MB write {write / 1e6:.1f} MB = {(fetch + write) / alg:.2f} x algorithmic ({alg / 1e6:.1f} MB)", flush=True)
    json.dump({"note": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE (separate passes) on tools/gemm_one.py, one MI355X, ROCm 7.2; FETCH_SIZE "
                       "(KB) doubled per MI355X_MICROARCH.md (gfx950 tallies 128-B requests at 64 B); WRITE_SIZE (KB) as reported. L2-miss (fabric) "
                       "traffic: Infinity-Cache hits are included. Split-K launches: the GEMM kernel only (partials written, not the reduce).",
               "command": "python tools/pmc_gemm_traffic.py", "kernels": res}, open(out_path, "w"), indent=1)


if __name__ == "__main__":
    main()
