#!/bin/bash
# Round 5: two-barrier flush (PEGAINFER_SKINNY_FLUSH=0) against the default, same box, alternating: $1 = model, $2 = batch.
cd ${GRAFT_REPO_ROOT:-/root/repo}; mkdir -p gpurun_out
m=${1:-qwen3-4b}; b=${2:-8}; ab=gpurun_out/r5_batch_flush_ab_${m}_bs$b.txt; : > $ab
for rep in 1 2; do for v in PEGAINFER_SKINNY_FLUSH=0 X=0; do
  env $v timeout 200 python bench.py --model $m --batch $b --steps 48 --cpu-steps 0 --ttft-iters 1 --profile-iters 0 --ttft10k-iters 0 --sweep-steps 0 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$m $v bs', d['config']['batch_per_gpu'], 'tok/s', d['value'], 'device_ms', d['tpot_ms']['device_p50'])" >> $ab
done; done; cat $ab
