#!/bin/bash
# round 3, call B: overlap feasibility probe + GEMV grid alignment A/B (same box)
repo=${GRAFT_REPO_ROOT:-/root/repo}
cd $repo
mkdir -p gpurun_out
timeout 120 tools/probes/overlap_probe 12 2>&1 | tee gpurun_out/r3b_overlap_probe.txt
ab() {  # label, env...
  label=$1; shift
  env "$@" timeout 200 python bench.py --steps 64 --cpu-steps 0 --ttft-iters 1 --profile-iters 0 --ttft10k-iters 0 2>/dev/null | tail -1 | \
    python -c "import sys,json;d=json.loads(sys.stdin.read());print('$label','tok/s',d['value'],'device_ms',d['tpot_ms']['device_p50'],{k:v['us'] for k,v in d['gemv_sites'].items()},'fused gate_up',d['roofline']['avg_launch_us'])"
}
for i in 1 2; do
  ab "mult0(old)      " PEGAINFER_GEMV_GRID_MULT=0
  ab "mult256         " PEGAINFER_GEMV_GRID_MULT=256
  ab "mult256 percu3  " PEGAINFER_GEMV_GRID_MULT=256 PEGAINFER_GEMV_GRID_PER_CU=3
done 2>&1 | tee gpurun_out/r3b_grid_ab.txt
