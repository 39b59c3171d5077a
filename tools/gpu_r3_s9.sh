#!/bin/bash
# round-3 session 9: attention kernels without the per-element mask branch / with vector softmax arithmetic: parity, kernel A/B against the
# previous library build (valor_amd/libvalor_hip_prev.so, VALOR_HIP_LIB), in-step A/B
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_attention_gpu.py tests/test_model_gpu.py -q > gpurun_out/s9_pytest.log 2>&1; echo "pytest rc=$?"; tail -6 gpurun_out/s9_pytest.log | cut -c1-300
VALOR_HIP_LIB=$R/valor_amd/libvalor_hip_prev.so timeout 300 python tools/kernels_one.py time gpurun_out/s9_kernels_prev.json 2>&1 | tail -1 | cut -c1-600
timeout 300 python tools/kernels_one.py time gpurun_out/s9_kernels_new.json 2>&1 | tail -1 | cut -c1-600
b() { n=$1; shift; env "$@" timeout 300 python bench.py --steps 10 --warmup 4 --no-cpu-baseline --sim-world 0 > gpurun_out/s9_bench_$n.log 2>&1; echo "$n: $(tail -1 gpurun_out/s9_bench_$n.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])" 2>&1 | tail -1)"; }
timeout 200 python bench.py --steps 3 --warmup 2 --no-cpu-baseline --sim-world 0 --no-roofline > /dev/null 2>&1
b prev VALOR_HIP_LIB=$R/valor_amd/libvalor_hip_prev.so
b new A=1
b prev2 VALOR_HIP_LIB=$R/valor_amd/libvalor_hip_prev.so
b new2 A=1
