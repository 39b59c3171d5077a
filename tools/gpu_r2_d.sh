#!/bin/bash
# round 2, session D: GPU tests + smoke, GEMM traffic PMC passes, headline bench + kernel trace, the other bench variants
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q -s > gpurun_out/pytest.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed|error" gpurun_out/pytest.log | tail -3
grep -E "^FAILED|^ERROR|bf16 vs reference|^E  " gpurun_out/pytest.log | cut -c1-420 | head -30
timeout 400 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; echo "smoke rc=$?"; grep "\[smoke\]" gpurun_out/smoke.log | cut -c1-200
timeout 900 python tools/pmc_gemm_traffic.py gpurun_out/r02_pmc_gemm_traffic.json > gpurun_out/pmc.log 2>&1; echo "pmc rc=$?"; cat gpurun_out/pmc.log | cut -c1-200
timeout 300 python bench.py > gpurun_out/bench.log 2>&1; echo "bench rc=$?"; tail -1 gpurun_out/bench.log | cut -c1-200
cp gpurun_out/bench.log gpurun_out/r02_bench_b64_v2.json
for v in swin large clip_large; do
  timeout 400 python bench.py --variant $v --steps 4 --warmup 2 --no-cpu-baseline > gpurun_out/bench_$v.log 2>&1; echo "bench $v rc=$?"; tail -1 gpurun_out/bench_$v.log | cut -c1-200
done
timeout 400 python bench.py --variant large --frames 16 --steps 3 --warmup 2 --no-cpu-baseline > gpurun_out/bench_large_f16.log 2>&1; echo "bench large f16 rc=$?"; tail -1 gpurun_out/bench_large_f16.log | cut -c1-200
cd /tmp; export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof -o r02 -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $R/gpurun_out/prof.log 2>&1; echo "prof rc=$?"
cd $R
DB=$(find gpurun_out/prof -name '*.db' | head -1)
[ -n "$DB" ] && python tools/rocpd_stats.py $DB gpurun_out/kernel_stats.md 45 | head -34 | cut -c1-130
find gpurun_out/prof -name '*.db' -delete
