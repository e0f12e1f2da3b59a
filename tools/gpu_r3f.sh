#!/bin/bash
# round 3, call F: GEMV whole-row-in-flight (U = 5) A/B + workgroups-per-CU sweep, then the fused / real-dims tests
repo=${GRAFT_REPO_ROOT:-/root/repo}
cd $repo
mkdir -p gpurun_out
ab() {  # label, env...
  label=$1; shift
  env "$@" timeout 200 python bench.py --steps 64 --cpu-steps 0 --ttft-iters 1 --profile-iters 0 --ttft10k-iters 0 2>/dev/null | tail -1 | \
    python -c "import sys,json;d=json.loads(sys.stdin.read());print('$label','tok/s',d['value'],'device_ms',d['tpot_ms']['device_p50'],{k:v['us'] for k,v in d['gemv_sites'].items()},'fused gate_up',d['roofline']['avg_launch_us'])"
}
for i in 1 2; do
  ab "u5=0          " PEGAINFER_GEMV_U5=0
  ab "u5=1          " PEGAINFER_GEMV_U5=1
  ab "u5=1 percu3   " PEGAINFER_GEMV_U5=1 PEGAINFER_GEMV_GRID_PER_CU=3
  ab "u5=1 percu2   " PEGAINFER_GEMV_U5=1 PEGAINFER_GEMV_GRID_PER_CU=2
done 2>&1 | tee gpurun_out/r3f_u5_ab.txt
timeout 600 python -m pytest tests/test_gpu_fused.py tests/test_gpu_real_dims.py tests/test_gpu_model.py -m gpu -q -x --tb=short > gpurun_out/pytest_r3f.log 2>&1
tail -3 gpurun_out/pytest_r3f.log
