#!/bin/bash
# round 3, call H: one row group per workgroup for the short GEMVs (RPW picked by shape) A/B + tests + phase trace
repo=${GRAFT_REPO_ROOT:-/root/repo}
cd $repo
mkdir -p gpurun_out
ab() {  # label, env...
  label=$1; shift
  env "$@" timeout 200 python bench.py --steps 64 --cpu-steps 0 --ttft-iters 1 --profile-iters 0 --ttft10k-iters 0 2>/dev/null | tail -1 | \
    python -c "import sys,json;d=json.loads(sys.stdin.read());print('$label','tok/s',d['value'],'device_ms',d['tpot_ms']['device_p50'],{k:v['us'] for k,v in d['gemv_sites'].items()},'fused gate_up',d['roofline']['avg_launch_us'])"
}
for i in 1 2; do
  ab "rpw_pick=0    " PEGAINFER_GEMV_RPW_PICK=0
  ab "rpw_pick=1    " PEGAINFER_GEMV_RPW_PICK=1
done 2>&1 | tee gpurun_out/r3h_rpw_ab.txt
timeout 200 python tools/gemv_probe.py --sites 6 1 3 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r3h_gemv_phase_trace.txt
timeout 900 python -m pytest tests/test_gpu_fused.py tests/test_gpu_real_dims.py tests/test_gpu_model.py tests/test_gpu_ops.py tests/test_gpu_real_dims_cfg34.py -m gpu -q -x --tb=short > gpurun_out/pytest_r3h.log 2>&1
tail -3 gpurun_out/pytest_r3h.log
