#!/bin/bash
repo=${GRAFT_REPO_ROOT:-/root/repo}
cd $repo
bash tools/gpu_tests.sh
for x in 1 0 1 0; do
  for c in 1024 2048; do
    PEGAINFER_GEMM256_TAIL=$x timeout 300 python bench.py --ctx $c --steps 4 --cpu-steps 0 --ttft-iters 4 2>/dev/null | tail -1 | python -c "import sys,json;d=json.loads(sys.stdin.read());print('tail $x ctx',$c,d.get('ttft_ms'))"
  done
done
