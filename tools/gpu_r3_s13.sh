#!/bin/bash
# round-3 session 13: LayerNorm kernels with a quarter / eighth wave per row (VideoSwin stage-1 widths): parity, the VideoSwin bench line;
# the default bench line with monotone logits buffers (device allocations in the timed region)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_kernel_variants_gpu.py tests/test_gemm_ln_gpu.py tests/test_swin_gpu.py tests/test_embed_ops_gpu.py -q > gpurun_out/s13_pytest.log 2>&1; echo "pytest rc=$?"; tail -4 gpurun_out/s13_pytest.log | cut -c1-300
timeout 600 python -m pytest tests/test_model_gpu.py -q -k "swin" > gpurun_out/s13_pytest_swin_model.log 2>&1; echo "pytest swin model rc=$?"; tail -2 gpurun_out/s13_pytest_swin_model.log | cut -c1-300
timeout 400 python bench.py --variant swin --steps 6 --warmup 3 --no-cpu-baseline --sim-world 0 > gpurun_out/s13_bench_swin.log 2>&1; echo "swin: $(tail -1 gpurun_out/s13_bench_swin.log | cut -c1-200)"
VALOR_LN_VARIANT=0 timeout 400 python bench.py --variant swin --steps 6 --warmup 3 --no-cpu-baseline --sim-world 0 > gpurun_out/s13_bench_swin_lnv0.log 2>&1; echo "swin one-wave-per-row LN: $(tail -1 gpurun_out/s13_bench_swin_lnv0.log | cut -c1-200)"
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --sim-world 0 > gpurun_out/s13_bench_default.log 2>&1; tail -1 gpurun_out/s13_bench_default.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['timed_region'])"
