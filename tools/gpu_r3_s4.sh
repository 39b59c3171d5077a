#!/bin/bash
# round-3 session 4: full GPU suite on the final defaults, bench, one-stream kernel trace, PMC traffic of the dominant GEMM kernels
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/s4_pytest.log 2>&1; echo "pytest rc=$?"; tail -12 gpurun_out/s4_pytest.log | cut -c1-300
b() { n=$1; shift; env "$@" timeout 300 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --sim-world 0 > gpurun_out/s4_bench_$n.log 2>&1; echo "$n: $(tail -1 gpurun_out/s4_bench_$n.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['peak_mem_gb'], d['reserved_mem_gb'], d['timed_region']['device_allocations'], d['roofline']['all_gemms'])" 2>&1 | tail -1)"; }
b default A=1
b fused4 VALOR_GEMM_FUSED3=0
b default2 A=1
timeout 300 python tools/pmc_gemm_traffic.py gpurun_out/r03_pmc_gemm_traffic.json 2>&1 | grep -v amdgpu.ids | tail -6
cd /tmp; export TMPDIR=/tmp
VALOR_ENCODER_STREAMS=0 timeout 400 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/s4_prof1s -o r03 -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --sim-world 0 > $R/gpurun_out/s4_prof1s.log 2>&1; echo "prof one-stream rc=$?"
cd $R
DB=$(find gpurun_out/s4_prof1s -name '*.db' | head -1); [ -n "$DB" ] && python tools/rocpd_stats.py $DB gpurun_out/s4_kernel_stats_1s.md 48 | head -40 | cut -c1-140
find gpurun_out -name '*.db' -size +30M -delete
