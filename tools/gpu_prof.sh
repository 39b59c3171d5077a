#!/bin/bash
# bench + rocprofv3 kernel trace only (no pytest)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out
timeout 300 python bench.py --no-cpu-baseline > gpurun_out/bench.log 2>&1; echo "bench rc=$?"; tail -1 gpurun_out/bench.log | cut -c1-160
cd /tmp; export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof -o r01 -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $R/gpurun_out/prof.log 2>&1; echo "prof rc=$?"
cd $R
DB=$(find gpurun_out/prof -name '*.db' | head -1)
[ -n "$DB" ] && python tools/rocpd_stats.py $DB gpurun_out/kernel_stats.md 45 | head -36 | cut -c1-120
find gpurun_out/prof -name '*.db' -size +40M -delete
