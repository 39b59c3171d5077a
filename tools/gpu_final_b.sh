#!/bin/bash
# final round, part B: the two bench lines (with cpu_baseline) and their rocprofv3 kernel traces
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out
timeout 400 python bench.py > gpurun_out/bench.log 2>&1; echo "bench rc=$?"; tail -1 gpurun_out/bench.log | cut -c1-260
timeout 400 python bench.py --variant swin > gpurun_out/bench_swin.log 2>&1; echo "bench swin rc=$?"; tail -1 gpurun_out/bench_swin.log | cut -c1-260
cd /tmp; export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof -o r01 -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $R/gpurun_out/prof.log 2>&1; echo "prof rc=$?"
timeout 400 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_swin -o r01s -- python $R/bench.py --variant swin --steps 3 --warmup 1 --no-cpu-baseline > $R/gpurun_out/prof_swin.log 2>&1; echo "prof swin rc=$?"
cd $R
DB=$(find gpurun_out/prof -name '*.db' | head -1); [ -n "$DB" ] && python tools/rocpd_stats.py $DB gpurun_out/kernel_stats.md 45 | head -14 | cut -c1-130
DB=$(find gpurun_out/prof_swin -name '*.db' | head -1); [ -n "$DB" ] && python tools/rocpd_stats.py $DB gpurun_out/kernel_stats_swin.md 45 | head -14 | cut -c1-130
find gpurun_out/prof gpurun_out/prof_swin -name '*.db' -size +40M -delete
