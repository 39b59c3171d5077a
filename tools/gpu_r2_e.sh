#!/bin/bash
# round 2, session E: GPU tests, GEMM traffic PMC passes, in-step A/B of the tile epilogue, large variants
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/pytest.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed|error" gpurun_out/pytest.log | tail -3
grep -E "^FAILED|^ERROR|^E  " gpurun_out/pytest.log | cut -c1-300 | head -20
timeout 900 python tools/pmc_gemm_traffic.py gpurun_out/r02_pmc_gemm_traffic.json > gpurun_out/pmc.log 2>&1; echo "pmc rc=$?"; cat gpurun_out/pmc.log | cut -c1-250 | head -30
for i in 1 2; do
  timeout 200 python bench.py --no-cpu-baseline > gpurun_out/bench_a$i.log 2>&1; echo "bench default: $(tail -1 gpurun_out/bench_a$i.log | cut -c1-150)"
  VALOR_GEMM_FAST_EPI=0 timeout 200 python bench.py --no-cpu-baseline > gpurun_out/bench_b$i.log 2>&1; echo "bench general epilogue: $(tail -1 gpurun_out/bench_b$i.log | cut -c1-150)"
done
for v in large clip_large; do
  timeout 400 python bench.py --variant $v --steps 4 --warmup 2 --no-cpu-baseline > gpurun_out/bench_$v.log 2>&1; echo "bench $v rc=$?"; tail -1 gpurun_out/bench_$v.log | cut -c1-250
done
timeout 400 python bench.py --variant large --frames 16 --steps 3 --warmup 2 --no-cpu-baseline > gpurun_out/bench_large_f16.log 2>&1; echo "bench large f16 rc=$?"; tail -1 gpurun_out/bench_large_f16.log | cut -c1-250
