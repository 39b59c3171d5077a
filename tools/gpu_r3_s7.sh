#!/bin/bash
# round-3 session 7: optimizer tail on the side stream (parity + A/B), full suite
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/s7_pytest.log 2>&1; echo "pytest rc=$?"; tail -6 gpurun_out/s7_pytest.log | cut -c1-300
b() { n=$1; shift; env "$@" timeout 300 python bench.py --steps 10 --warmup 4 --no-cpu-baseline --sim-world 0 > gpurun_out/s7_bench_$n.log 2>&1; echo "$n: $(tail -1 gpurun_out/s7_bench_$n.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['timed_region']['device_allocations'], d['roofline']['all_gemms'])" 2>&1 | tail -1)"; }
b split A=1
b nosplit VALOR_ADAMW_SPLIT=0
b split2 A=1
b nosplit2 VALOR_ADAMW_SPLIT=0
