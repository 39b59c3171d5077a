#!/bin/bash
# VideoSwin variant, one-stream rocprofv3 kernel trace (3 timed steps after 2 warm-up steps) -> profiles-style summary + GEMM launches by grid
R=${GRAFT_REPO_ROOT:-/root/repo}; cd /tmp; export TMPDIR=/tmp
TAG=${1:-v1}
export VALOR_ENCODER_STREAMS=0 VALOR_KV_STREAM=0
rm -rf $R/gpurun_out/prof_swin4
timeout 400 rocprofv3 --kernel-trace -d $R/gpurun_out/prof_swin4 -o t -- python $R/bench.py --variant swin --steps 3 --warmup 2 --no-cpu-baseline --no-roofline --no-variants --sim-world 0 > $R/gpurun_out/prof_swin4.log 2>&1; echo "prof rc=$?"
tail -1 $R/gpurun_out/prof_swin4.log | cut -c1-300
DB=$(find $R/gpurun_out/prof_swin4 -name '*.db' | head -1)
python $R/tools/rocpd_stats.py $DB $R/gpurun_out/r04_swin_b64_kernel_stats_$TAG.md 60 | head -50 | cut -c1-150
python $R/tools/rocpd_gemm_by_grid.py $DB 5 > $R/gpurun_out/r04_swin_gemm_by_grid_$TAG.txt
find $R/gpurun_out/prof_swin4 -name '*.db' -delete
