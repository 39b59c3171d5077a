#!/bin/bash
# round-3 session 11: per-kernel timing of the non-GEMM hot kernels (previous library build vs this one), the small GEMM shapes against the
# vendor library, bench with the slack-allocated logits buffers (device allocations in the timed region)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out
VALOR_HIP_LIB=$R/valor_amd/libvalor_hip_prev.so timeout 300 python tools/kernels_one.py time gpurun_out/s11_kernels_prev.json > gpurun_out/s11_kernels_prev.log 2>&1; tail -1 gpurun_out/s11_kernels_prev.log | cut -c1-700
timeout 300 python tools/kernels_one.py time gpurun_out/s11_kernels_new.json > gpurun_out/s11_kernels_new.log 2>&1; tail -1 gpurun_out/s11_kernels_new.log | cut -c1-700
timeout 300 python tools/gemm_vs_library.py gpurun_out/s11_gemm_small.json small > gpurun_out/s11_gemm_small.log 2>&1; tail -24 gpurun_out/s11_gemm_small.log | cut -c1-220
b() { n=$1; shift; env "$@" timeout 300 python bench.py --steps 10 --warmup 4 --no-cpu-baseline --sim-world 0 > gpurun_out/s11_bench_$n.log 2>&1; echo "$n: $(tail -1 gpurun_out/s11_bench_$n.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['timed_region'])" 2>&1 | tail -1)"; }
b a A=1
b b A=1
