#!/bin/bash
# round 5, session 2: family-4 GEMM on v_mfma_f32_32x32x16_bf16 + the two-output tile epilogue: parity, then the A/B
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gemm_narrow_gpu.py -m gpu -q > gpurun_out/pytest_s2.log 2>&1; echo "pytest rc=$?"; tail -8 gpurun_out/pytest_s2.log
timeout 600 python tools/gemm_mfma32_ab.py gpurun_out/r05_gemm_mfma32_ab.json > gpurun_out/mfma32_ab.log 2>&1; echo "ab rc=$?"; cat gpurun_out/mfma32_ab.log | cut -c1-900
timeout 300 python -m pytest tests/test_model_gpu.py -m gpu -q -k "f16a2 and bf16" > gpurun_out/pytest_s2b.log 2>&1; echo "pytest2 rc=$?"; grep "bf16 vs reference\|passed\|failed" gpurun_out/pytest_s2b.log | cut -c1-600
