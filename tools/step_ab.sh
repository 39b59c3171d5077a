#!/bin/bash
# in-step A/B of environment presets: tools/step_ab.sh OUT "NAME1:ENV=.. ENV=.." "NAME2:..." ; each runs bench.py (no CPU baseline, no roofline) and prints samples/s
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; mkdir -p gpurun_out
OUT=$1; shift
: > gpurun_out/$OUT
for rep in 1 2; do
for spec in "$@"; do
  name=${spec%%:*}; envs=${spec#*:}
  v=$(env $envs timeout 300 python bench.py --no-cpu-baseline --no-roofline --steps ${STEPS:-10} --warmup 4 ${BENCH_ARGS} 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])")
  echo "$rep $name [$envs] $v" | tee -a gpurun_out/$OUT
done; done
