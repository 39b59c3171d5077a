#!/bin/bash
# round-3 session 16 (final state): full GPU suite, smoke(), the bench line with the driver's flags, the VideoSwin line
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/s16_pytest.log 2>&1; echo "pytest rc=$?"; tail -4 gpurun_out/s16_pytest.log | cut -c1-200
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/s16_smoke.log 2>&1; echo "smoke rc=$?"; tail -1 gpurun_out/s16_smoke.log
timeout 400 python bench.py --variant swin --steps 6 --warmup 3 --no-cpu-baseline --sim-world 0 > gpurun_out/s16_bench_swin.log 2>&1; echo "swin: $(tail -1 gpurun_out/s16_bench_swin.log | cut -c1-200)"
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/s16_bench_driver_flags.log 2>&1; echo "bench rc=$?"; tail -1 gpurun_out/s16_bench_driver_flags.log | cut -c1-400
