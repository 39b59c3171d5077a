#!/bin/bash
# round-3 session 15: window-attention dQ kernel without SGPR spills (opaque chunk offset) and with the bounds select only in the last chunk:
# parity (tests/test_swin_gpu.py + the swin model goldens), kernel time (tools/win_one.py), in-step A/B against the previous library build
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_swin_gpu.py -q > gpurun_out/s15_pytest.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/s15_pytest.log | cut -c1-300
timeout 600 python -m pytest tests/test_model_gpu.py -q -k "swin" > gpurun_out/s15_pytest_swin_model.log 2>&1; echo "pytest swin model rc=$?"; tail -2 gpurun_out/s15_pytest_swin_model.log | cut -c1-300
b() { n=$1; shift; env "$@" timeout 400 python bench.py --variant swin --steps 6 --warmup 3 --no-cpu-baseline --sim-world 0 > gpurun_out/s15_bench_swin_$n.log 2>&1; echo "$n: $(tail -1 gpurun_out/s15_bench_swin_$n.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['losses'])" 2>&1 | tail -1)"; }
b prev VALOR_HIP_LIB=$R/valor_amd/libvalor_hip_prev.so
b new A=1
b prev2 VALOR_HIP_LIB=$R/valor_amd/libvalor_hip_prev.so
b new2 A=1
