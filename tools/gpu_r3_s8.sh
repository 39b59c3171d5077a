#!/bin/bash
# round-3 session 8: single-pass attention backward (mode 2): parity, kernel A/B, in-step A/B
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_attention_gpu.py -q > gpurun_out/s8_pytest_attn.log 2>&1; echo "pytest rc=$?"; tail -8 gpurun_out/s8_pytest_attn.log | cut -c1-300
timeout 300 python tools/attn_pipe_ab.py gpurun_out/s8_attn_pipe_ab.json 2>&1 | tail -4 | cut -c1-400
b() { n=$1; shift; env "$@" timeout 300 python bench.py --steps 10 --warmup 4 --no-cpu-baseline --sim-world 0 > gpurun_out/s8_bench_$n.log 2>&1; echo "$n: $(tail -1 gpurun_out/s8_bench_$n.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])" 2>&1 | tail -1)"; }
b pipe1 VALOR_ATTN_PIPE=1
b pass1 VALOR_ATTN_PIPE=2
b pipe1b VALOR_ATTN_PIPE=1
b pass1b VALOR_ATTN_PIPE=2
