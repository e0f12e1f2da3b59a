#!/bin/bash
# Run on the GPU box: bench.py decode device ms for a list of "args" strings (one per line in AB_ARGS, ';'-separated)
repo=${GRAFT_REPO_ROOT:-/root/repo}
cd $repo
IFS=';' read -ra runs <<< "${AB_ARGS:---batch 16;--batch 64;--batch 1}"
for r in "${runs[@]}"; do
  timeout 200 python bench.py $r --steps 48 --cpu-steps 0 --ttft-iters 1 --profile-iters 0 --ttft10k-iters 0 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$r', '->', d['value'], 'tok/s device_ms', d['tpot_ms']['device_p50'], 'ttft', d['ttft_ms']['p50'])"
done
