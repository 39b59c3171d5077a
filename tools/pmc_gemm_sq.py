"""Aggregate the rocprofv3 --pmc CSVs of tools/pmc_gemm_sq.sh: per policy (n0 = base, n1 = narrow family 4) the counters of the LAST gemm dispatch."""
import csv
import glob
import os
import sys

root = sys.argv[1]
for nar in (0, 1):
    vals, name, dur = {}, None, None
    for d in sorted(glob.glob(os.path.join(root, f"n{nar}_p*"))):
        if not os.path.isdir(d):
            continue
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            rows = [r for r in csv.DictReader(open(f)) if "gemm" in r.get("Kernel_Name", "")]
            if not rows:
                continue
            last = max(int(r["Dispatch_Id"]) for r in rows)
            for r in rows:
                if int(r["Dispatch_Id"]) == last:
                    name = r["Kernel_Name"]
                    vals[r["Counter_Name"]] = vals.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
        for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
            rows = [r for r in csv.DictReader(open(f)) if "gemm" in r.get("Kernel_Name", "")]
            if rows:
                r = rows[-1]
                dur = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    print(f"## narrow={nar}: `{(name or '?')[:80]}`  last dispatch {dur} us (profiled pass)")
    wc = max(vals.get("SQ_WAVE_CYCLES", 0.0), 1.0)
    for k in sorted(vals):
        print(f"- {k} = {vals[k]:.5g}" + (f"  ({100 * vals[k] / wc:.1f} % of SQ_WAVE_CYCLES)" if k.startswith("SQ_") and k not in ("SQ_WAVE_CYCLES", "SQ_WAVES") and "INSTS" not in k and "LEVEL" not in k and "MFMA_BUSY" not in k else ""))
    if dur and "SQ_VALU_MFMA_BUSY_CYCLES" in vals:
        ga = vals.get("GRBM_GUI_ACTIVE")
        print(f"- MFMA pipe busy (sum over 1024 SIMDs) / (dur x 2.1 GHz x 1024) = {vals['SQ_VALU_MFMA_BUSY_CYCLES'] / (dur * 2100.0 * 1024.0):.3f}" + (f"; GRBM_GUI_ACTIVE/dur = {ga / dur / 1e3:.3f} GHz (sum over XCDs / 8 if > 3)" if ga else ""))
    print()
