#!/bin/bash
# One GPU-box session: parity tests, bench, GEMM microbench, host profile, rocprofv3 kernel trace.
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out
timeout 1200 python -m pytest tests -m gpu -x -q > gpurun_out/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/pytest.log
timeout 400 python bench.py > gpurun_out/bench.log 2>&1; echo "bench rc=$?"; tail -1 gpurun_out/bench.log
timeout 200 python tools/bench_gemm.py gpurun_out/gemm.json > gpurun_out/gemm.log 2>&1; grep bfloat16 gpurun_out/gemm.log
timeout 200 python tools/host_profile.py > gpurun_out/host.log 2>&1; head -5 gpurun_out/host.log
cd /tmp; export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof -o r01 -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $R/gpurun_out/prof.log 2>&1; echo "prof rc=$?"
cd $R
DB=$(find gpurun_out/prof -name '*.db' | head -1); echo "db=$DB"
[ -n "$DB" ] && python tools/rocpd_stats.py $DB gpurun_out/kernel_stats.md 45 | head -40
find gpurun_out/prof -name '*.db' -size +40M -delete
