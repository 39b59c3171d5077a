#!/bin/bash
# round-end validation on one GPU box: the default bench line FIRST (cold box, as the driver runs it), smoke(), then the full GPU suite
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; mkdir -p gpurun_out
TAG=${1:-final}
timeout 600 python bench.py > gpurun_out/r04_bench_default_$TAG.log 2>&1; echo "bench rc=$?"; tail -1 gpurun_out/r04_bench_default_$TAG.log > gpurun_out/r04_bench_default_$TAG.json; cut -c1-400 gpurun_out/r04_bench_default_$TAG.json
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/r04_smoke_$TAG.txt 2>&1; echo "smoke rc=$?"; tail -2 gpurun_out/r04_smoke_$TAG.txt
timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/r04_pytest_gpu_$TAG.txt 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r04_pytest_gpu_$TAG.txt
