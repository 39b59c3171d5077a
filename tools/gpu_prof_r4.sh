#!/bin/bash
# one-stream rocprofv3 kernel trace of the bench step (per-kernel durations are only meaningful without concurrency) -> profiles-style summary
R=${GRAFT_REPO_ROOT:-/root/repo}; cd /tmp; export TMPDIR=/tmp
TAG=${1:-v1}
export VALOR_ENCODER_STREAMS=0 VALOR_KV_STREAM=0
rm -rf $R/gpurun_out/prof_r4
timeout 400 rocprofv3 --kernel-trace -d $R/gpurun_out/prof_r4 -o t -- python $R/bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-roofline --no-variants --sim-world 0 > $R/gpurun_out/prof_r4.log 2>&1; echo "prof rc=$?"
DB=$(find $R/gpurun_out/prof_r4 -name '*.db' | head -1)
python $R/tools/rocpd_stats.py $DB $R/gpurun_out/r04_bench_b64_kernel_stats_$TAG.md 70 | head -45 | cut -c1-150
python $R/tools/rocpd_gemm_by_grid.py $DB 10 > $R/gpurun_out/r04_gemm_by_grid_$TAG.txt
find $R/gpurun_out/prof_r4 -name '*.db' -delete
