#!/bin/bash
# round 3, call J: in-register row merge in decode attention: same-box A/B against the _ab tree, phase trace, tests
repo=${GRAFT_REPO_ROOT:-/root/repo}
cd $repo
mkdir -p gpurun_out
bash tools/gpu_ab_tree.sh "1 2 16" 64 2>&1 | tee gpurun_out/r3j_rowmerge_ab.txt
timeout 200 python tools/attn_probe.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r3j_attn_phase_trace.txt
timeout 200 python tools/attn_probe.py --batch 16 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/r3j_attn_phase_trace.txt
timeout 900 python -m pytest tests/test_gpu_fused.py tests/test_gpu_real_dims.py tests/test_gpu_model.py tests/test_gpu_ops.py tests/test_gpu_qwen35.py tests/test_gpu_qwen35_model.py -m gpu -q -x --tb=short > gpurun_out/pytest_r3j.log 2>&1
tail -5 gpurun_out/pytest_r3j.log
