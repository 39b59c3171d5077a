#!/bin/bash
# round-3 session 10: full GPU suite on the current tree (attention <DROP, MASK> kernels, gradient sinks of the embedding / assembly /
# conv-as-GEMM parameters), per-kernel timing of the non-GEMM hot kernels against the previous library build, in-step A/B
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/s10_pytest.log 2>&1; echo "pytest rc=$?"; tail -8 gpurun_out/s10_pytest.log | cut -c1-300
VALOR_HIP_LIB=$R/valor_amd/libvalor_hip_prev.so timeout 300 python tools/kernels_one.py time gpurun_out/s10_kernels_prev.json > gpurun_out/s10_kernels_prev.log 2>&1; tail -3 gpurun_out/s10_kernels_prev.log | cut -c1-700
timeout 300 python tools/kernels_one.py time gpurun_out/s10_kernels_new.json > gpurun_out/s10_kernels_new.log 2>&1; tail -3 gpurun_out/s10_kernels_new.log | cut -c1-700
b() { n=$1; shift; env "$@" timeout 300 python bench.py --steps 10 --warmup 4 --no-cpu-baseline --sim-world 0 > gpurun_out/s10_bench_$n.log 2>&1; echo "$n: $(tail -1 gpurun_out/s10_bench_$n.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])" 2>&1 | tail -1)"; grep -c "Warning" gpurun_out/s10_bench_$n.log; }
b new A=1
b new2 A=1
