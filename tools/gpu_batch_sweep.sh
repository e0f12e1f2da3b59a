#!/bin/bash
# Run on the GPU box (via gpurun): batch sweep @ctx 1024 (+ optional serving points) into gpurun_out/.
repo=${GRAFT_REPO_ROOT:-/root/repo}
cd $repo
mkdir -p gpurun_out
for b in ${SWEEP_BATCHES:-2 4 8 16 32 64}; do
  timeout 200 python bench.py --batch $b --steps 48 --cpu-steps 0 --ttft-iters 1 --profile-iters 0 --ttft10k-iters 0 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('bs', d['config']['batch_per_gpu'], 'tok/s', d['value'], 'device_ms', d['tpot_ms']['device_p50'])"
done > gpurun_out/r1_batch_sweep.txt
cat gpurun_out/r1_batch_sweep.txt
for c in ${SWEEP_CONC:-8 32}; do
  timeout 250 python bench.py --concurrency $c --steps 128 --cpu-steps 0 2>/dev/null | tail -1 > gpurun_out/serving_c$c.json
  python -c "import json; d=json.load(open('gpurun_out/serving_c$c.json')); print('serving C=$c', d['value'], 'tok/s')"
done
