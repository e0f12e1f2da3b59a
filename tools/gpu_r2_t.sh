#!/bin/bash
repo=${GRAFT_REPO_ROOT:-/root/repo}
cd $repo
sw() { # dir label
  for b in 1 16; do
    (cd $1 && timeout 200 python bench.py --batch $b --steps 96 --cpu-steps 0 --ttft-iters 1 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$2 bs', d['config']['batch_per_gpu'], 'tok/s', d['value'], 'device_ms', d['tpot_ms']['device_p50'])")
  done
}
sw _ab head
sw . new
sw _ab head
sw . new
