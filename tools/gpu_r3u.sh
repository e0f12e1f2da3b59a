#!/bin/bash
repo=${GRAFT_REPO_ROOT:-/root/repo}
cd $repo
ab() {  # label, args, env...
  label=$1; shift; args=$1; shift
  env "$@" timeout 200 python bench.py $args --steps 64 --cpu-steps 0 --ttft-iters 1 --profile-iters 0 --ttft10k-iters 0 2>/dev/null | tail -1 | \
    python -c "import sys,json;d=json.loads(sys.stdin.read());print('$label','tok/s',d['value'],'device_ms',d['tpot_ms']['device_p50'],{k:v['us'] for k,v in d['gemv_sites'].items()})"
}
for i in 1 2; do
  ab "4b cap>=40KB (new default)" "" PEGAINFER_X=0
  ab "4b cap all (percu2)       " "" PEGAINFER_GEMV_GRID_PER_CU=2
  ab "4b cap>=16KB              " "" PEGAINFER_GEMV_CAP_MIN_KB=16
  ab "8b cap>=40KB (new default)" "--model qwen3-8b" PEGAINFER_X=0
  ab "8b cap>=32KB              " "--model qwen3-8b" PEGAINFER_GEMV_CAP_MIN_KB=32
  ab "q35 new default" "--model qwen3.5-4b" PEGAINFER_X=0 2>/dev/null
done 2>&1 | tee gpurun_out/r3u_cap_rule_ab.txt
