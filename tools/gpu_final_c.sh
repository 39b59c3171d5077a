#!/bin/bash
# re-measure after the asm transposing reads (GEMM) and the LDS-DMA dQ pass became the defaults
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_kernel_variants_gpu.py tests/test_swin_gpu.py tests/test_gemm_ln_gpu.py -q -x 2>&1 | tail -2
timeout 400 python bench.py > gpurun_out/bench.log 2>&1; echo "bench rc=$?"; tail -1 gpurun_out/bench.log | cut -c1-200
timeout 400 python bench.py --variant swin --no-cpu-baseline > gpurun_out/bench_swin.log 2>&1; echo "bench swin rc=$?"; tail -1 gpurun_out/bench_swin.log | cut -c1-200
cd /tmp; export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof -o r01 -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $R/gpurun_out/prof.log 2>&1; echo "prof rc=$?"
timeout 400 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_swin -o r01s -- python $R/bench.py --variant swin --steps 3 --warmup 1 --no-cpu-baseline > $R/gpurun_out/prof_swin.log 2>&1; echo "prof swin rc=$?"
cd $R
DB=$(find gpurun_out/prof -name '*.db' | head -1); [ -n "$DB" ] && python tools/rocpd_stats.py $DB gpurun_out/kernel_stats.md 45 | head -12 | cut -c1-130
DB=$(find gpurun_out/prof_swin -name '*.db' | head -1); [ -n "$DB" ] && python tools/rocpd_stats.py $DB gpurun_out/kernel_stats_swin.md 45 | head -12 | cut -c1-130
find gpurun_out/prof gpurun_out/prof_swin -name '*.db' -size +40M -delete
