#!/bin/bash
# round 3, call G: where a step's time is now - kernel trace of the default bench, GEMV + attention phase traces
repo=${GRAFT_REPO_ROOT:-/root/repo}
cd $repo
mkdir -p gpurun_out
bash tools/gpu_kt.sh r3g_decode --steps 48 --cpu-steps 0 --ttft-iters 1 --profile-iters 0 --ttft10k-iters 0 2>&1 | head -30
timeout 200 python tools/gemv_probe.py --sites 6 1 5 3 7 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r3g_gemv_phase_trace.txt
timeout 200 python tools/attn_probe.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r3g_attn_phase_trace.txt
