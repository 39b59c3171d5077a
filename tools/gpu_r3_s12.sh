#!/bin/bash
# round-3 session 12 (closing): full GPU suite, smoke(), the default bench line (cpu_baseline + dp_sim), one-stream kernel trace, per-kernel
# counter table, the VideoSwin variant (bench line + kernel trace)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q --durations=6 > gpurun_out/s12_pytest.log 2>&1; echo "pytest rc=$?"; tail -12 gpurun_out/s12_pytest.log | cut -c1-200
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/s12_smoke.log 2>&1; echo "smoke rc=$?"; tail -2 gpurun_out/s12_smoke.log
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/s12_bench_driver_flags.log 2>&1; echo "bench rc=$?"; tail -1 gpurun_out/s12_bench_driver_flags.log | cut -c1-2500
cd /tmp; export TMPDIR=/tmp
VALOR_ENCODER_STREAMS=0 timeout 400 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/s12_prof1s -o r03 -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --sim-world 0 > $R/gpurun_out/s12_prof1s.log 2>&1; echo "prof one-stream rc=$?"
timeout 400 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/s12_prof_swin -o r03s -- python $R/bench.py --variant swin --steps 3 --warmup 1 --no-cpu-baseline --sim-world 0 > $R/gpurun_out/s12_prof_swin.log 2>&1; echo "prof swin rc=$?"
cd $R
DB=$(find gpurun_out/s12_prof1s -name '*.db' | head -1); [ -n "$DB" ] && python tools/rocpd_stats.py $DB gpurun_out/s12_kernel_stats_1s.md 50 > /dev/null
DB=$(find gpurun_out/s12_prof_swin -name '*.db' | head -1); [ -n "$DB" ] && python tools/rocpd_stats.py $DB gpurun_out/s12_kernel_stats_swin.md 45 > /dev/null
find gpurun_out -name '*.db' -size +30M -delete
bash tools/gpu_pmc_kernels.sh 2>&1 | tail -30 | cut -c1-250
timeout 400 python bench.py --variant swin --steps 6 --warmup 3 --no-cpu-baseline --sim-world 0 > gpurun_out/s12_bench_swin.log 2>&1; echo "swin: $(tail -1 gpurun_out/s12_bench_swin.log | cut -c1-260)"
timeout 300 python bench.py --no-cpu-baseline --sim-world 0 > gpurun_out/s12_bench_default2.log 2>&1; tail -1 gpurun_out/s12_bench_default2.log | cut -c1-400
