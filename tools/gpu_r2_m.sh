#!/bin/bash
# session M: derivative-saving activation pair (ACT_DERIV): GEMM tests, shapes A/B, model parity subset, step A/B vs the previous build
cd ${GRAFT_REPO_ROOT:-/root/repo}; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gemm_ln_gpu.py tests/test_kernel_variants_gpu.py -q -m gpu -x > gpurun_out/pytest_m.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/pytest_m.log
timeout 900 python -m pytest tests/test_model_gpu.py -q -m gpu -x -k "tiny or north_star or goldens" > gpurun_out/pytest_m2.log 2>&1; echo "pytest2 rc=$?"; tail -3 gpurun_out/pytest_m2.log
timeout 300 python tools/gemm_policy_ab.py gpurun_out/r02_gemm_epilogue_ab_v4_deriv.json > gpurun_out/gemm_deriv.log 2>&1; echo "ab rc=$?"; grep -E "fc1_fwd|dgrad_d|fc2_dgrad " gpurun_out/gemm_deriv.log | cut -c1-200
for i in 1 2; do
  VALOR_MLP_DERIV=0 timeout 200 python bench.py --no-cpu-baseline > gpurun_out/ab_pre_$i.log 2>&1; echo "keeps x  : $(tail -1 gpurun_out/ab_pre_$i.log | cut -c50-150)"
  timeout 200 python bench.py --no-cpu-baseline > gpurun_out/ab_deriv_$i.log 2>&1; echo "keeps act': $(tail -1 gpurun_out/ab_deriv_$i.log | cut -c50-150)"
done
