#!/bin/bash
# round-3 session 18: split-K reduce with 16-byte loads / four K-slices in flight: parity, in-step A/B against the previous library build
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out
timeout 300 python -m pytest tests/test_gemm_bench_shapes_gpu.py tests/test_kernel_variants_gpu.py -q -k "wgrad or splitk or gemm_families or logits" > gpurun_out/s18_pytest.log 2>&1; echo "pytest rc=$?"; tail -2 gpurun_out/s18_pytest.log | cut -c1-200
b() { n=$1; shift; env "$@" timeout 200 python bench.py --steps 10 --warmup 4 --no-cpu-baseline --sim-world 0 --no-roofline > gpurun_out/s18_bench_$n.log 2>&1; echo "$n: $(tail -1 gpurun_out/s18_bench_$n.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['losses']['total_loss'])" 2>&1 | tail -1)"; }
b prev VALOR_HIP_LIB=$R/valor_amd/libvalor_hip_prev.so
b new A=1
b prev2 VALOR_HIP_LIB=$R/valor_amd/libvalor_hip_prev.so
b new2 A=1
