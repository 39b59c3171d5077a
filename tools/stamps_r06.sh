cd $GRAFT_REPO_ROOT
for v in stamp abl1 abl2 abl3 abl4; do
  for w in 1 0; do
    [ $w = 0 ] && [ $v = abl4 ] && continue
    WIDE=$w VALOR_HIP_LIB=valor_amd/libvalor_hip_$v.so python tools/gemm_stamp.py 100864 3072 768 0 0 gpurun_out/r06_stamp_fc1fwd_${v}_wide$w.json > /dev/null 2>&1
    python - <<PY
import json
d=json.load(open("gpurun_out/r06_stamp_fc1fwd_${v}_wide$w.json"))
print("$v wide=$w", d["kernel_us_with_stamps"], "us  prologue", d["prologue"]["median"], "kloop", d["k_loop"]["median"], "per-tile", round(d["k_loop_per_tile_median"]), "epi", d["epilogue"]["median"], "total", d["total"]["median"], "ge1", round(d["frac_span_with_ge1_wg_in_k_loop"],3), "both", round(d["frac_span_with_2_wg_in_k_loop"],3))
PY
  done
done
