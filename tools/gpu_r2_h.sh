#!/bin/bash
# round 2, session H: new dropout hash + cross-attention trims (tests), cross-attention register experiments (launch bounds), A/B on the bench
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_attention_gpu.py tests/test_kernel_variants_gpu.py -m gpu -q > gpurun_out/pytest_attn.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed|error" gpurun_out/pytest_attn.log | tail -3
grep -E "^FAILED|^ERROR|^E  " gpurun_out/pytest_attn.log | cut -c1-300 | head -10
for L in prev "" lb1 lb2; do
  if [ -n "$L" ]; then export VALOR_HIP_LIB=$R/valor_amd/libvalor_hip_$L.so; else unset VALOR_HIP_LIB; fi
  echo "== lib ${L:-tree}"; timeout 300 python tools/attn_x_ab.py 3 2>&1 | grep -E "FAIL|caption|mlm" | cut -c1-200
done
unset VALOR_HIP_LIB
timeout 300 python tools/attn_ab.py 3 2>&1 | grep -E "FAIL|vit|ast|dec_self|mlm_self" | cut -c1-200
bash tools/ab_bench.sh valor_amd/libvalor_hip_prev.so 2
