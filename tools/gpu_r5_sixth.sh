#!/bin/bash
# Round 5, sixth GPU call: does zeroing the in-launch counters with a KERNEL node instead of a memset node end the
# "one eager launch between two graph replays breaks every later replay" failure?  Then the sampling-sensitive tests.
repo=${GRAFT_REPO_ROOT:-/root/repo}
cd $repo
mkdir -p gpurun_out
timeout 900 python tools/diag_8b.py > gpurun_out/r5_diag_8b_4.txt 2>&1
timeout 900 python -m pytest tests/test_gpu_full_depth_8b.py tests/test_gpu_model.py tests/test_gpu_logprobs.py tests/test_gpu_fused.py -m gpu -q > gpurun_out/r5_zero_kernel_tests.log 2>&1
echo "pytest rc $?" >> gpurun_out/r5_zero_kernel_tests.log
grep -v amdgpu gpurun_out/r5_diag_8b_4.txt
grep -E "passed|failed|rc |Error" gpurun_out/r5_zero_kernel_tests.log | tail -6
