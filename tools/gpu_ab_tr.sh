#!/bin/bash
# A/B: inline-asm transposing reads (8-phase GEMM) and the LDS-DMA dQ pass of the window attention
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out
VALOR_GEMM_TR_ASM=1 timeout 600 python -m pytest tests/test_kernel_variants_gpu.py tests/test_gemm_ln_gpu.py -q -x 2>&1 | tail -2
VALOR_WIN_VARIANT=1 timeout 300 python -m pytest tests/test_swin_gpu.py -q -x -k window_attention 2>&1 | tail -2
for V in 0 1; do echo "win variant $V"; VALOR_WIN_VARIANT=$V python tools/win_one.py 2 3 2>&1 | tail -2; VALOR_WIN_VARIANT=$V python tools/win_one.py 0 3 2>&1 | tail -2; done
for V in 0 1; do VALOR_GEMM_TR_ASM=$V timeout 300 python bench.py --no-cpu-baseline > gpurun_out/bench_tr$V.log 2>&1; echo "tr_asm=$V rc=$?"; tail -1 gpurun_out/bench_tr$V.log | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print(d['value'], d['ms_per_step'], {k.split(' ')[0][:14] + k.split(')')[1][:4]: (v['TFLOPs'], v['avg_us']) for k, v in d['roofline']['all_gemm_kernels'].items()})"; done
