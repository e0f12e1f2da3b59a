#!/bin/bash
# full suite, then TTFT with an env knob off / on alternating: tools/gpu_ttft_ab.sh PEGAINFER_SPLITK256 "512 1024 2048"
repo=${GRAFT_REPO_ROOT:-/root/repo}
knob=$1; ctxs=${2:-"1024"}
cd $repo
bash tools/gpu_tests.sh
for x in 1 0 1 0; do
  for c in $ctxs; do
    env $knob=$x timeout 300 python bench.py --ctx $c --steps 4 --cpu-steps 0 --ttft-iters 4 --profile-iters 0 --ttft10k-iters 0 2>/dev/null | tail -1 | python -c "import sys,json;d=json.loads(sys.stdin.read());print('$knob=$x ctx',$c,d.get('ttft_ms'))"
  done
done
