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
  ab "off               " "" PEGAINFER_ATTN_OPROJ=0
  ab "on  chunks18 h0   " "" PEGAINFER_ATTN_OPROJ_HOLD=0
  ab "on  chunks12 h0   " "" PEGAINFER_ATTN_OPROJ_HOLD=0 PEGAINFER_OPROJ_CHUNKS=12
  ab "on  chunks9  h0   " "" PEGAINFER_ATTN_OPROJ_HOLD=0 PEGAINFER_OPROJ_CHUNKS=9
  ab "off chunks9       " "" PEGAINFER_ATTN_OPROJ=0 PEGAINFER_OPROJ_CHUNKS=9
  ab "on  chunks9  h160 " "" PEGAINFER_OPROJ_CHUNKS=9
done 2>&1 | tee gpurun_out/r3r_attn_oproj_ab.txt
ab "on chunks9 ctx 2048" "--ctx 2048" PEGAINFER_ATTN_OPROJ_HOLD=0 PEGAINFER_OPROJ_CHUNKS=9 | tee -a gpurun_out/r3r_attn_oproj_ab.txt
ab "on chunks18 ctx 2048" "--ctx 2048" PEGAINFER_ATTN_OPROJ_HOLD=0 | tee -a gpurun_out/r3r_attn_oproj_ab.txt
