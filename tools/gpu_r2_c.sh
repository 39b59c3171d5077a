#!/bin/bash
# round 2, session C: GPU tests, smoke, attention family A/B (split backward), GEMM / LN policies re-checked in the step, bench + kernel trace
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q -s > gpurun_out/pytest.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed|error" gpurun_out/pytest.log | tail -3
grep -E "^FAILED|^ERROR|bf16 vs reference|^E  " gpurun_out/pytest.log | cut -c1-420 | head -30
timeout 400 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; echo "smoke rc=$?"; grep "\[smoke\]" gpurun_out/smoke.log | cut -c1-200
timeout 400 python tools/attn_ab.py 0,3,7 > gpurun_out/attn_ab.log 2>&1; echo "attn rc=$?"; grep -E "FAIL|dropout|vit|ast|dec_self|mlm_self" gpurun_out/attn_ab.log | cut -c1-330
timeout 300 python bench.py --no-cpu-baseline > gpurun_out/bench.log 2>&1; echo "bench rc=$?"; tail -1 gpurun_out/bench.log | cut -c1-200
cd /tmp; export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof -o r02 -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $R/gpurun_out/prof.log 2>&1; echo "prof rc=$?"
cd $R
DB=$(find gpurun_out/prof -name '*.db' | head -1)
[ -n "$DB" ] && python tools/rocpd_stats.py $DB gpurun_out/kernel_stats.md 45 | head -40 | cut -c1-130
find gpurun_out/prof -name '*.db' -size +40M -delete
