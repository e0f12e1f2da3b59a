#!/bin/bash
# Round 5: the whole -m gpu suite with per-test durations, then the default bench line (what the driver runs at round end).
repo=${GRAFT_REPO_ROOT:-/root/repo}
cd $repo
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q --durations=25 > gpurun_out/r5_pytest_gpu_full.log 2>&1
echo "pytest rc $?" >> gpurun_out/r5_pytest_gpu_full.log
timeout 600 python bench.py --steps 20 --warmup 5 > gpurun_out/r5_bench_driver_style.json 2> gpurun_out/r5_bench_driver_style.err
echo "bench rc $?" >> gpurun_out/r5_bench_driver_style.err
grep -E "passed|failed|error|rc " gpurun_out/r5_pytest_gpu_full.log | tail -8
grep -A30 "slowest" gpurun_out/r5_pytest_gpu_full.log | head -32
tail -c 900 gpurun_out/r5_bench_driver_style.json
