#!/bin/bash
# round 3, call I: tagged-granule partial hand-off in decode attention: A/B, phase trace, bit-identity tests
repo=${GRAFT_REPO_ROOT:-/root/repo}
cd $repo
mkdir -p gpurun_out
ab() {  # label, args, env...
  label=$1; shift; args=$1; shift
  env "$@" timeout 200 python bench.py $args --steps 64 --cpu-steps 0 --ttft-iters 1 --profile-iters 0 --ttft10k-iters 0 2>/dev/null | tail -1 | \
    python -c "import sys,json;d=json.loads(sys.stdin.read());print('$label','tok/s',d['value'],'device_ms',d['tpot_ms']['device_p50'])"
}
for i in 1 2; do
  ab "granules=0 bs1" "" PEGAINFER_ATTN_GRANULES=0
  ab "granules=1 bs1" "" PEGAINFER_ATTN_GRANULES=1
  ab "granules=0 bs2" "--batch 2" PEGAINFER_ATTN_GRANULES=0
  ab "granules=1 bs2" "--batch 2" PEGAINFER_ATTN_GRANULES=1
done 2>&1 | tee gpurun_out/r3i_granule_ab.txt
ab "granules=1 ctx 8192" "--ctx 8192" PEGAINFER_ATTN_GRANULES=1 | tee -a gpurun_out/r3i_granule_ab.txt
ab "granules=0 ctx 8192" "--ctx 8192" PEGAINFER_ATTN_GRANULES=0 | tee -a gpurun_out/r3i_granule_ab.txt
timeout 200 python tools/attn_probe.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r3i_attn_phase_trace.txt
timeout 900 python -m pytest tests/test_gpu_fused.py tests/test_gpu_real_dims.py tests/test_gpu_model.py -m gpu -q -x --tb=short > gpurun_out/pytest_r3i.log 2>&1
tail -5 gpurun_out/pytest_r3i.log
