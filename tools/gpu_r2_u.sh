#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}; mkdir -p gpurun_out
timeout 200 python bench.py > gpurun_out/bench_final2.log 2>&1; echo "bench rc=$?"; tail -1 gpurun_out/bench_final2.log | python -c "
import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']
print(d['value'], d['ms_per_step'], 'frac', r['frac'], r['kernel'], r['achieved'], 'all', r['all_gemms'], 'mfu', r['step_mfu'], d['timed_region'])
for k,v in r['all_gemm_kernels'].items(): print(k, v)"
