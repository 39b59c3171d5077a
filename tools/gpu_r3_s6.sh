#!/bin/bash
# round-3 session 6: full GPU suite, smoke(), the default bench line (cpu_baseline + dp_sim), one-stream kernel trace, per-kernel PMC table,
# the other variants
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q --durations=8 > gpurun_out/s6_pytest.log 2>&1; echo "pytest rc=$?"; tail -14 gpurun_out/s6_pytest.log | cut -c1-200
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/s6_smoke.log 2>&1; echo "smoke rc=$?"; tail -2 gpurun_out/s6_smoke.log
timeout 600 python bench.py > gpurun_out/s6_bench_default.log 2>&1; echo "bench rc=$?"; tail -1 gpurun_out/s6_bench_default.log | cut -c1-3000
cd /tmp; export TMPDIR=/tmp
VALOR_ENCODER_STREAMS=0 timeout 400 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/s6_prof1s -o r03 -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --sim-world 0 > $R/gpurun_out/s6_prof1s.log 2>&1; echo "prof one-stream rc=$?"
cd $R
DB=$(find gpurun_out/s6_prof1s -name '*.db' | head -1); [ -n "$DB" ] && python tools/rocpd_stats.py $DB gpurun_out/s6_kernel_stats_1s.md 50 > /dev/null
find gpurun_out -name '*.db' -size +30M -delete
bash tools/gpu_pmc_kernels.sh 2>&1 | tail -50 | cut -c1-250
for v in swin large clip_large; do timeout 400 python bench.py --variant $v --steps 4 --warmup 2 --no-cpu-baseline --sim-world 0 > gpurun_out/s6_bench_$v.log 2>&1; echo "$v: $(tail -1 gpurun_out/s6_bench_$v.log | cut -c1-260)"; done
timeout 300 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --sim-world 0 > gpurun_out/s6_bench_default2.log 2>&1; tail -1 gpurun_out/s6_bench_default2.log | cut -c1-200
