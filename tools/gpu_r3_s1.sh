#!/bin/bash
# round-3 session 1: new parity tests, GEMM L2 knob sweep (timing + PMC), bench with the new defaults and A/B of each
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gemm_bench_shapes_gpu.py tests/test_contrastive_fused_gpu.py tests/test_contrastive_xent_gpu.py "tests/test_model_gpu.py::test_bf16_meets_north_star_on_identical_tensors" tests/test_dp_model_gpu.py::test_bench_gpus_flag_launches_that_many_ranks -x -q -s > gpurun_out/s1_pytest.log 2>&1; echo "pytest rc=$?"; tail -5 gpurun_out/s1_pytest.log | cut -c1-400
grep "bf16 vs reference" gpurun_out/s1_pytest.log | cut -c1-500
timeout 300 python tools/gemm_l2_ab.py time gpurun_out/s1_gemm_l2_ab.json > gpurun_out/s1_gemm_l2_ab.log 2>&1; echo "l2ab rc=$?"; cat gpurun_out/s1_gemm_l2_ab.log | cut -c1-420
cd /tmp; export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $R/gpurun_out/s1_pmc_fetch -o p -- python $R/tools/gemm_l2_ab.py pmc $R/gpurun_out/s1_pmc_order.json > $R/gpurun_out/s1_pmc_fetch.log 2>&1; echo "pmc fetch rc=$?"
timeout 200 rocprofv3 --kernel-trace --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum -d $R/gpurun_out/s1_pmc_write -o p -- python $R/tools/gemm_l2_ab.py pmc $R/gpurun_out/s1_pmc_order2.json > $R/gpurun_out/s1_pmc_write.log 2>&1; echo "pmc write rc=$?"
cd $R
F=$(find gpurun_out/s1_pmc_fetch -name '*.db' | head -1); W=$(find gpurun_out/s1_pmc_write -name '*.db' | head -1)
python tools/gemm_l2_pmc.py gpurun_out/s1_pmc_order.json gpurun_out/s1_pmc_join.json FETCH_SIZE=$F WRITE_SIZE=$W TCC_HIT_sum=$W TCC_MISS_sum=$W 2>&1 | cut -c1-330
find gpurun_out -name '*.db' -size +30M -delete
timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/s1_bench_default.log 2>&1; echo "bench rc=$?"; tail -1 gpurun_out/s1_bench_default.log | cut -c1-1500
VALOR_GEMM_RASTER=0 timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --sim-world 0 > gpurun_out/s1_bench_raster0.log 2>&1; tail -1 gpurun_out/s1_bench_raster0.log | cut -c1-200
VALOR_GEMM_FUSED3=0 timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --sim-world 0 > gpurun_out/s1_bench_fused4.log 2>&1; tail -1 gpurun_out/s1_bench_fused4.log | cut -c1-200
VALOR_FINE_FUSED=0 timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/s1_bench_unfused_contra.log 2>&1; tail -1 gpurun_out/s1_bench_unfused_contra.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d.get('dp_sim'))"
timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --sim-world 0 > gpurun_out/s1_bench_default2.log 2>&1; tail -1 gpurun_out/s1_bench_default2.log | cut -c1-200
