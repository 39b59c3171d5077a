cd $GRAFT_REPO_ROOT
for v in ${STAMP_LIBS:-stamp}; do
  for w in ${STAMP_WIDE:-1 0}; do
    WIDE=$w VALOR_HIP_LIB=valor_amd/libvalor_hip_$v.so python tools/gemm_stamp.py ${STAMP_SHAPE:-100864 3072 768 0 0} gpurun_out/r06_stamp_${STAMP_TAG:-fc1fwd}_${v}_wide$w.json > /dev/null 2>&1
    python - <<PY
import json
d=json.load(open("gpurun_out/r06_stamp_${STAMP_TAG:-fc1fwd}_${v}_wide$w.json"))
print("$v wide=$w", d["kernel_us_with_stamps"], "us  prologue", d["prologue"]["median"], "kloop", d["k_loop"]["median"], "per-tile", round(d["k_loop_per_tile_median"]), "epi", d["epilogue"]["median"], "total", d["total"]["median"], "ge1", round(d["frac_span_with_ge1_wg_in_k_loop"],3), "both", round(d["frac_span_with_2_wg_in_k_loop"],3))
print("   ", d.get("k_tile_5_segments_median"), d.get("k_tile_5_total_median"))
PY
  done
done
