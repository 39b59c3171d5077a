#!/bin/bash
# session S: why is the first default bench run of a session slow? device allocations / host time inside the timed region
cd ${GRAFT_REPO_ROOT:-/root/repo}; mkdir -p gpurun_out
for i in 1 2; do
timeout 200 python bench.py --no-cpu-baseline > gpurun_out/s_default_$i.log 2>&1; tail -1 gpurun_out/s_default_$i.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('default    ', d['value'], d['ms_per_step'], d['reserved_mem_gb'], d['timed_region'])"
done
timeout 200 python bench.py --no-cpu-baseline --steps 20 > gpurun_out/s_20.log 2>&1; tail -1 gpurun_out/s_20.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('steps 20   ', d['value'], d['ms_per_step'], d['reserved_mem_gb'], d['timed_region'])"
VALOR_ENCODER_STREAMS=0 timeout 200 python bench.py --no-cpu-baseline --steps 20 > gpurun_out/s_20_1s.log 2>&1; tail -1 gpurun_out/s_20_1s.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('1s steps 20', d['value'], d['ms_per_step'], d['reserved_mem_gb'], d['timed_region'])"
