#!/bin/bash
# round 2, session B: GPU tests (large config, tile epilogue), bf16 error attribution, GEMM epilogue A/B, LN family A/B, bench
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q -s > gpurun_out/pytest.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed|error" gpurun_out/pytest.log | tail -3
grep -E "^FAILED|^ERROR|bf16 vs reference|^E  " gpurun_out/pytest.log | cut -c1-500 | head -30
timeout 400 python tools/bf16_error_attribution.py ref_base_b2f2a1_q gpurun_out/r02_bf16_attribution_b2f2a1.json > gpurun_out/attr.log 2>&1; echo "attr rc=$?"; grep -v Warning gpurun_out/attr.log | cut -c1-1800
timeout 300 python tools/ln_bench.py gpurun_out/r02_ln_ab.json > gpurun_out/ln.log 2>&1; echo "ln rc=$?"; tail -8 gpurun_out/ln.log | cut -c1-300
timeout 400 python tools/gemm_policy_ab.py gpurun_out/r02_gemm_epilogue_ab.json > gpurun_out/gemm_policy.log 2>&1; echo "policy rc=$?"; grep -v amdgpu.ids gpurun_out/gemm_policy.log | cut -c1-300
timeout 300 python bench.py --no-cpu-baseline > gpurun_out/bench.log 2>&1; echo "bench rc=$?"; tail -1 gpurun_out/bench.log | cut -c1-200
