#!/bin/bash
# round 2, session A: full GPU test suite (new bf16 / F=8 / DP / contrastive cases), smoke, GEMM policy A/B, bench
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q -s > gpurun_out/pytest.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed|error" gpurun_out/pytest.log | tail -3
grep -E "^FAILED|^ERROR|bf16 vs reference" gpurun_out/pytest.log | cut -c1-600
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; echo "smoke rc=$?"; grep "\[smoke\]" gpurun_out/smoke.log | cut -c1-200
timeout 500 python tools/gemm_policy_ab.py gpurun_out/r02_gemm_policy_ab.json > gpurun_out/gemm_policy.log 2>&1; echo "policy rc=$?"; cat gpurun_out/gemm_policy.log | cut -c1-400
timeout 300 python bench.py --no-cpu-baseline > gpurun_out/bench.log 2>&1; echo "bench rc=$?"; tail -1 gpurun_out/bench.log | cut -c1-300
