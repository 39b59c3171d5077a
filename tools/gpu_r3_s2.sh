#!/bin/bash
# round-3 session 2: templated nt-store GEMM epilogues (parity + timing + PMC), LayerNorm / AdamW nt A/B, library ceiling, contrastive profile, bench A/B
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gemm_bench_shapes_gpu.py tests/test_gemm_ln_gpu.py "tests/test_kernel_variants_gpu.py" -x -q > gpurun_out/s2_pytest.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/s2_pytest.log | cut -c1-300
timeout 200 python tools/gemm_l2_ab.py time gpurun_out/s2_gemm_l2_ab.json 2>&1 | grep -v amdgpu.ids | cut -c1-400
timeout 200 python tools/stream_nt_ab.py gpurun_out/s2_stream_nt_ab.json 2>&1 | grep -v amdgpu.ids | cut -c1-600
timeout 300 python tools/gemm_vs_library.py gpurun_out/s2_gemm_vs_library.json 2>&1 | grep -v amdgpu.ids | cut -c1-300
cd /tmp; export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $R/gpurun_out/s2_pmc_fetch -o p -- python $R/tools/gemm_l2_ab.py pmc $R/gpurun_out/s2_pmc_order.json > $R/gpurun_out/s2_pmc_fetch.log 2>&1; echo "pmc fetch rc=$?"
timeout 200 rocprofv3 --kernel-trace --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum -d $R/gpurun_out/s2_pmc_write -o p -- python $R/tools/gemm_l2_ab.py pmc $R/gpurun_out/s2_pmc_order2.json > $R/gpurun_out/s2_pmc_write.log 2>&1; echo "pmc write rc=$?"
timeout 200 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/s2_contra -o c -- python $R/tools/contra_prof.py 512 4 > $R/gpurun_out/s2_contra.log 2>&1; echo "contra prof rc=$?"
cd $R
F=$(find gpurun_out/s2_pmc_fetch -name '*.db' | head -1); W=$(find gpurun_out/s2_pmc_write -name '*.db' | head -1)
python tools/gemm_l2_pmc.py gpurun_out/s2_pmc_order.json gpurun_out/s2_pmc_join.json FETCH_SIZE=$F WRITE_SIZE=$W TCC_HIT_sum=$W TCC_MISS_sum=$W 2>&1 | cut -c1-330
C=$(find gpurun_out/s2_contra -name '*.db' | head -1); python tools/rocpd_stats.py $C gpurun_out/s2_contra_kernel_stats.md 30 | cut -c1-160
find gpurun_out -name '*.db' -size +30M -delete
timeout 300 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --sim-world 0 > gpurun_out/s2_bench_default.log 2>&1; echo "bench rc=$?"; tail -1 gpurun_out/s2_bench_default.log | cut -c1-220
VALOR_GEMM_STORE=0 timeout 300 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --sim-world 0 > gpurun_out/s2_bench_store0.log 2>&1; tail -1 gpurun_out/s2_bench_store0.log | cut -c1-220
VALOR_GEMM_NTA=1 timeout 300 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --sim-world 0 > gpurun_out/s2_bench_nts128.log 2>&1; tail -1 gpurun_out/s2_bench_nts128.log | cut -c1-220
VALOR_LN_NT=31 VALOR_ADAMW_NT=3 timeout 300 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --sim-world 0 > gpurun_out/s2_bench_lnnt.log 2>&1; tail -1 gpurun_out/s2_bench_lnnt.log | cut -c1-220
timeout 300 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --sim-world 0 > gpurun_out/s2_bench_default2.log 2>&1; tail -1 gpurun_out/s2_bench_default2.log | cut -c1-220
