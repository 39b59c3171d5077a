#!/bin/bash
# round-3 session 14: the 8-phase GEMM at K = 128 / 256 (VideoSwin stages 1 / 2): parity + time against the 128 x 128 kernels, in-step A/B
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out
timeout 300 python tools/gemm_smallk_check.py gpurun_out/s14_gemm_smallk.json > gpurun_out/s14_gemm_smallk.log 2>&1; echo "check rc=$?"; tail -8 gpurun_out/s14_gemm_smallk.log | cut -c1-420
b() { n=$1; shift; env "$@" timeout 400 python bench.py --variant swin --steps 6 --warmup 3 --no-cpu-baseline --sim-world 0 > gpurun_out/s14_bench_swin_$n.log 2>&1; echo "$n: $(tail -1 gpurun_out/s14_bench_swin_$n.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['losses'])" 2>&1 | tail -1)"; }
b base A=1
b nn128 VALOR_GEMM_NN_MINK=128
b nn128_nt128 VALOR_GEMM_NN_MINK=128 VALOR_GEMM_NT_MINK=128
b nt128 VALOR_GEMM_NT_MINK=128
