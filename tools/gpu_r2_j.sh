#!/bin/bash
# session J: finetune tasks + generation, VideoSwin padding, activation rewrite (GEMM tests, A/B of the act shapes, step A/B vs the previous build)
cd ${GRAFT_REPO_ROOT:-/root/repo}; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_finetune_gpu.py tests/test_swin_gpu.py tests/test_gemm_ln_gpu.py tests/test_kernel_variants_gpu.py -x -q -m gpu > gpurun_out/pytest_j.log 2>&1; echo "pytest rc=$?"; tail -5 gpurun_out/pytest_j.log
timeout 600 python -m pytest tests/test_model_gpu.py -x -q -m gpu -k "tiny or north_star" > gpurun_out/pytest_j2.log 2>&1; echo "pytest2 rc=$?"; tail -3 gpurun_out/pytest_j2.log
for lib in valor_amd/libvalor_hip_prev.so valor_amd/libvalor_hip.so; do
  VALOR_HIP_LIB=$PWD/$lib timeout 300 python tools/gemm_policy_ab.py gpurun_out/gemm_act_$(basename $lib .so).json > gpurun_out/gemm_act_$(basename $lib .so).log 2>&1; echo "$lib rc=$?"
  grep -E "fc1_fwd |dact|ast_fc1|dec_fc" gpurun_out/gemm_act_$(basename $lib .so).log | head -12
done
bash tools/ab_bench.sh valor_amd/libvalor_hip_prev.so 2
