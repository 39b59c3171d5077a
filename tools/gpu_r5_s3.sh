#!/bin/bash
# round 5, session 3: device-resident dropout counter + hipGraph capture of the encoders (parity, host time, in-step A/B), the dropout /
# attention / LayerNorm kernel suites behind the ABI change, and the cycle stamps of the 16x16x32 vs 32x32x16 main loops
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_graphs_gpu.py -m gpu -q -x > gpurun_out/pytest_s3_graphs.log 2>&1; echo "graphs pytest rc=$?"; tail -15 gpurun_out/pytest_s3_graphs.log
timeout 900 python -m pytest tests/test_gemm_ln_gpu.py tests/test_attention_gpu.py tests/test_kernel_variants_gpu.py tests/test_cross_attn_fused_gpu.py tests/test_static_kv_gpu.py -m gpu -q > gpurun_out/pytest_s3_kernels.log 2>&1; echo "kernels pytest rc=$?"; tail -4 gpurun_out/pytest_s3_kernels.log
for g in 0 1; do VALOR_GRAPHS=$g timeout 300 python tools/host_profile.py > gpurun_out/host_graphs$g.log 2>&1; echo "host graphs=$g rc=$?"; head -3 gpurun_out/host_graphs$g.log; done
STEPS=10 bash tools/step_ab.sh r05_step_ab_s3_graphs.txt "eager:VALOR_GRAPHS=0" "graphs:VALOR_GRAPHS=1"
if [ -f valor_amd/libvalor_hip_stamp.so ]; then
  for m in 0 1; do MFMA32=$m VALOR_HIP_LIB=valor_amd/libvalor_hip_stamp.so timeout 200 python tools/gemm_stamp.py 100864 3072 768 0 0 gpurun_out/r05_stamp_fc1fwd_mfma32_$m.json > gpurun_out/stamp_$m.log 2>&1; echo "stamp m32=$m rc=$?"; grep -A12 "k_tile_5_segments" gpurun_out/stamp_$m.log | tr -d '\n' | cut -c1-600; echo; grep "k_loop_per_tile_median\|k_tile_5_total" gpurun_out/stamp_$m.log; done
fi
