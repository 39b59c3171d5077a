#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}; mkdir -p gpurun_out
timeout 200 python bench.py --no-cpu-baseline --steps 20 > gpurun_out/t_20.log 2>&1; tail -1 gpurun_out/t_20.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('enc-only 20', d['value'], d['ms_per_step'], d['reserved_mem_gb'], d['timed_region'])"
timeout 200 python bench.py --no-cpu-baseline > gpurun_out/t_5.log 2>&1; tail -1 gpurun_out/t_5.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('enc-only 5 ', d['value'], d['ms_per_step'], d['reserved_mem_gb'], d['timed_region'])"
