#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_kernel_variants_gpu.py -m gpu -q -k "gemm" > gpurun_out/pytest_gemm.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed|error" gpurun_out/pytest_gemm.log | tail -3
grep -E "^FAILED|^ERROR|^E  " gpurun_out/pytest_gemm.log | cut -c1-300 | head -10
timeout 500 python tools/gemm_policy_ab.py gpurun_out/r02_gemm_epilogue_ab_v2.json > gpurun_out/gemm_policy.log 2>&1; echo "policy rc=$?"; grep -v amdgpu.ids gpurun_out/gemm_policy.log | cut -c1-330
for m in 1 3 4 1 3 4; do VALOR_GEMM_FAST_EPI=$m timeout 200 python bench.py --no-cpu-baseline > gpurun_out/bench_m$m.log 2>&1; echo "epilogue mode $m: $(tail -1 gpurun_out/bench_m$m.log | cut -c50-150)"; done
