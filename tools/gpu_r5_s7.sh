#!/bin/bash
# round 5, session 7: the two regressions found in the s6 trace (tile-epilogue register allocation, scalar RNG load inside the attention tile
# loops) fixed -- in-session A/B of the two builds, the touched kernel suites, and the full-depth large parity case
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out
STEPS=10 bash tools/step_ab.sh r05_step_ab_s7_fix.txt "s6_build:VALOR_HIP_LIB=valor_amd/libvalor_hip_prev.so" "fixed:VALOR_X=0" 2>&1 | tail -4
timeout 900 python -m pytest tests/test_gemm_narrow_gpu.py tests/test_attention_gpu.py tests/test_cross_attn_fused_gpu.py tests/test_graphs_gpu.py tests/test_kernel_variants_gpu.py -m gpu -q > gpurun_out/pytest_s7.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/pytest_s7.log
timeout 600 python -m pytest tests/test_model_gpu.py -m gpu -q -s -k "full_depth" > gpurun_out/pytest_s7_large.log 2>&1; echo "large rc=$?"; grep "bf16 vs oracle\|passed\|failed\|Error" gpurun_out/pytest_s7_large.log | cut -c1-700
