#!/bin/bash
# session N: dgrad (NT) min-tiles policy sweep on the step
cd ${GRAFT_REPO_ROOT:-/root/repo}; mkdir -p gpurun_out
for mt in 256 1024 512 256 1024; do
  VALOR_GEMM_NT_MIN_TILES=$mt timeout 200 python bench.py --no-cpu-baseline > gpurun_out/bench_nt$mt.log 2>&1; echo "NT min tiles $mt: $(tail -1 gpurun_out/bench_nt$mt.log | cut -c50-150)"
done
