#!/bin/bash
# session K: finetune/generation tests (bf16 teacher-forced), VideoSwin padding, GEMM/LN/variant tests on the new activations, min-tiles policy sweep
cd ${GRAFT_REPO_ROOT:-/root/repo}; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_finetune_gpu.py tests/test_swin_gpu.py tests/test_gemm_ln_gpu.py tests/test_kernel_variants_gpu.py -q -m gpu -s > gpurun_out/pytest_k.log 2>&1; echo "pytest rc=$?"; grep -E "bf16 max logit|passed|failed|^FAILED|^E  " gpurun_out/pytest_k.log | head -30
for mt in 1024 512 256 128 1024; do
  VALOR_GEMM_MIN_TILES=$mt timeout 200 python bench.py --no-cpu-baseline > gpurun_out/bench_mt$mt.log 2>&1; echo "min tiles $mt: $(tail -1 gpurun_out/bench_mt$mt.log | cut -c50-150)"
done
