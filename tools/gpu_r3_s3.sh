#!/bin/bash
# round-3 session 3: full GPU suite on the new defaults (pipelined attention backward, side-stream K|V projections with static buffers,
# fused contrastive v2), attention A/B, contrastive profile, bench A/B of every new default
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out
timeout 1200 python -m pytest tests -m gpu -x -q > gpurun_out/s3_pytest.log 2>&1; echo "pytest rc=$?"; tail -4 gpurun_out/s3_pytest.log | cut -c1-400
timeout 200 python tools/attn_pipe_ab.py gpurun_out/s3_attn_pipe_ab.json 2>&1 | grep -v amdgpu.ids | cut -c1-300
cd /tmp; export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/s3_contra -o c -- python $R/tools/contra_prof.py 512 4 > $R/gpurun_out/s3_contra.log 2>&1; echo "contra prof rc=$?"
cd $R
C=$(find gpurun_out/s3_contra -name '*.db' | head -1); python tools/rocpd_stats.py $C gpurun_out/s3_contra_kernel_stats.md 12 | cut -c1-150
find gpurun_out -name '*.db' -size +30M -delete
b() { n=$1; shift; env "$@" timeout 300 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --sim-world 0 > gpurun_out/s3_bench_$n.log 2>&1; echo "$n: $(tail -1 gpurun_out/s3_bench_$n.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['peak_mem_gb'], d['reserved_mem_gb'], d['timed_region']['device_allocations'])" 2>&1 | tail -1)"; }
b default A=1
b nopipe VALOR_ATTN_PIPE=0
b nokvstream VALOR_KV_STREAM=0
b splitk16 VALOR_GEMM_SPLITK_BF16=1
b raster VALOR_GEMM_RASTER=1000
b default2 A=1
timeout 300 python bench.py --steps 8 --warmup 2 --no-cpu-baseline > gpurun_out/s3_bench_dpsim.log 2>&1; tail -1 gpurun_out/s3_bench_dpsim.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d.get('dp_sim'))"
