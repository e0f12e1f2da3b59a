#!/bin/bash
repo=${GRAFT_REPO_ROOT:-/root/repo}
cd $repo
mkdir -p gpurun_out
ab() {  # label, args, env...
  label=$1; shift; args=$1; shift
  env "$@" timeout 200 python bench.py $args --steps 64 --cpu-steps 0 --ttft-iters 1 --profile-iters 0 --ttft10k-iters 0 2>/dev/null | tail -1 | \
    python -c "import sys,json;d=json.loads(sys.stdin.read());print('$label','tok/s',d['value'],'device_ms',d['tpot_ms']['device_p50'])"
}
for i in 1 2; do
  ab "attn_oproj=0      " "" PEGAINFER_ATTN_OPROJ=0
  ab "attn_oproj=1 late h160" "" PEGAINFER_ATTN_OPROJ=1
  ab "attn_oproj=1 late h0" "" PEGAINFER_ATTN_OPROJ=1 PEGAINFER_ATTN_OPROJ_HOLD=0
done 2>&1 | tee gpurun_out/r3o_attn_oproj_ab.txt
timeout 200 python tools/attn_probe.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r3o_attn_oproj_trace.txt
timeout 600 python -m pytest tests/test_gpu_fused.py tests/test_gpu_real_dims.py -m gpu -q -x --tb=short > gpurun_out/pytest_r3o.log 2>&1
tail -3 gpurun_out/pytest_r3o.log
