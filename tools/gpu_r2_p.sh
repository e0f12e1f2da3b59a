#!/bin/bash
repo=${GRAFT_REPO_ROOT:-/root/repo}
cd $repo
bash tools/gpu_tests.sh
for b in 8 16; do
  timeout 200 python bench.py --batch $b --steps 64 --cpu-steps 0 --ttft-iters 1 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('bs', d['config']['batch_per_gpu'], 'tok/s', d['value'], 'device_ms', d['tpot_ms']['device_p50'], d['gemv_sites']['down'])"
done
