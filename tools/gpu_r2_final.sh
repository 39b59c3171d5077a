#!/bin/bash
# round-2 final: full GPU suite, smoke(), the default bench line (cpu_baseline included), rocprofv3 kernel trace of one-stream steps (per-kernel
# durations are only meaningful without concurrency)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/pytest_final.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/pytest_final.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke_final.log 2>&1; echo "smoke rc=$?"; tail -2 gpurun_out/smoke_final.log
timeout 400 python bench.py > gpurun_out/bench_final.log 2>&1; echo "bench rc=$?"; tail -1 gpurun_out/bench_final.log
cd /tmp; export TMPDIR=/tmp
VALOR_ENCODER_STREAMS=0 timeout 400 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof1s -o r02 -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $R/gpurun_out/prof1s.log 2>&1; echo "prof one-stream rc=$?"
cd $R
DB=$(find gpurun_out/prof1s -name '*.db' | head -1); [ -n "$DB" ] && python tools/rocpd_stats.py $DB gpurun_out/kernel_stats_final_1s.md 45 | head -24 | cut -c1-130
find gpurun_out/prof1s -name '*.db' -size +40M -delete
