#!/bin/bash
# Run on the GPU box: the whole -m gpu suite, result kept in gpurun_out/pytest_gpu.log (gpurun returns only a tail)
repo=${GRAFT_REPO_ROOT:-/root/repo}
cd $repo
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q "$@" > gpurun_out/pytest_gpu.log 2>&1
echo "pytest rc $?" >> gpurun_out/pytest_gpu.log
grep -E "passed|failed|error|rc " gpurun_out/pytest_gpu.log | tail -5
