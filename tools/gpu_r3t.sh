#!/bin/bash
repo=${GRAFT_REPO_ROOT:-/root/repo}
cd $repo
ab() {  # label, args, env...
  label=$1; shift; args=$1; shift
  env "$@" timeout 200 python bench.py $args --steps 64 --cpu-steps 0 --ttft-iters 1 --profile-iters 0 --ttft10k-iters 0 2>/dev/null | tail -1 | \
    python -c "import sys,json;d=json.loads(sys.stdin.read());print('$label','tok/s',d['value'],'device_ms',d['tpot_ms']['device_p50'],{k:v['us'] for k,v in d['gemv_sites'].items()})"
}
for i in 1 2; do
  ab "8b percu2 (default)" "--model qwen3-8b" PEGAINFER_X=0
  ab "8b percu3" "--model qwen3-8b" PEGAINFER_GEMV_GRID_PER_CU=3
  ab "8b percu0 (occupancy)" "--model qwen3-8b" PEGAINFER_GEMV_GRID_PER_CU=0
  ab "8b percu0 mult0 (round 2)" "--model qwen3-8b" PEGAINFER_GEMV_GRID_PER_CU=0 PEGAINFER_GEMV_GRID_MULT=0
  ab "8b oproj off" "--model qwen3-8b" PEGAINFER_ATTN_OPROJ=0
done 2>&1 | tee gpurun_out/r3t_8b_ab.txt
