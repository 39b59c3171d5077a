#!/bin/bash
# round-3 session 5: epilogue operand prefetch + residual GradSlot (parity), attention backward variants, per-kernel PMC evidence, bench
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_attention_gpu.py tests/test_gemm_bench_shapes_gpu.py tests/test_kernel_variants_gpu.py tests/test_model_gpu.py tests/test_dp_model_gpu.py -q --durations=12 > gpurun_out/s5_pytest.log 2>&1; echo "pytest rc=$?"; tail -22 gpurun_out/s5_pytest.log | cut -c1-200
timeout 200 python tools/attn_pipe_ab.py gpurun_out/s5_attn_pipe_ab.json 2>&1 | grep -v amdgpu.ids | cut -c1-330
timeout 200 python tools/gemm_l2_ab.py time gpurun_out/s5_gemm_l2_ab.json 2>&1 | grep -v amdgpu.ids | cut -c1-300
bash tools/gpu_pmc_kernels.sh 2>&1 | tail -45
b() { n=$1; shift; env "$@" timeout 300 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --sim-world 0 > gpurun_out/s5_bench_$n.log 2>&1; echo "$n: $(tail -1 gpurun_out/s5_bench_$n.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['peak_mem_gb'], d['timed_region']['device_allocations'], d['roofline']['all_gemms'])" 2>&1 | tail -1)"; }
b default A=1
b swp VALOR_ATTN_PIPE=2
b default2 A=1
