#!/bin/bash
# round-6 validation on one GPU box: the bench line with the DRIVER S FLAGS first (cold box, first process: what the driver measures), smoke(), the full GPU suite, then the
# one-stream rocprofv3 kernel trace of the step (per-kernel durations are only meaningful without concurrency) and the GEMM PMC passes
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; mkdir -p gpurun_out
TAG=${1:-final}
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r06_bench_default_$TAG.log 2>&1; echo "bench rc=$?"; tail -1 gpurun_out/r06_bench_default_$TAG.log > gpurun_out/r06_bench_default_$TAG.json; cut -c1-300 gpurun_out/r06_bench_default_$TAG.json
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/r06_smoke_$TAG.txt 2>&1; echo "smoke rc=$?"; tail -2 gpurun_out/r06_smoke_$TAG.txt
timeout 1800 python -m pytest tests -m gpu -q > gpurun_out/r06_pytest_gpu_$TAG.txt 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r06_pytest_gpu_$TAG.txt
[ -n "$SKIP_PROF" ] && exit 0          # bench + smoke + suite only
cd /tmp; export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof_r6
VALOR_ENCODER_STREAMS=0 VALOR_KV_STREAM=0 timeout 400 rocprofv3 --kernel-trace -d $R/gpurun_out/prof_r6 -o t -- python $R/bench.py --steps 8 --warmup 2 --graphs 0 --no-cpu-baseline --no-roofline --no-variants --sim-world 0 > $R/gpurun_out/prof_r6.log 2>&1; echo "prof rc=$?"
DB=$(find $R/gpurun_out/prof_r6 -name '*.db' | head -1)
python $R/tools/rocpd_stats.py $DB $R/gpurun_out/r06_bench_b64_kernel_stats_$TAG.md 70 | head -30 | cut -c1-150
python $R/tools/rocpd_gemm_by_grid.py $DB 10 > $R/gpurun_out/r06_gemm_by_grid_$TAG.txt
find $R/gpurun_out/prof_r6 -name '*.db' -delete
cd $R
timeout 600 python tools/pmc_gemm_traffic.py gpurun_out/r06_pmc_gemm_traffic.json > gpurun_out/pmc_r6.log 2>&1; echo "pmc rc=$?"; tail -3 gpurun_out/pmc_r6.log | cut -c1-300
rm -rf gpurun_out/pmc
