#!/bin/bash
# In-pipeline A/B of the round-3 prefill changes: TTFT at 512 / 1024 / 4096 / 10 000 tokens with each one switched off
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
run() {  # label, env...
  local label=$1; shift
  for c in 512 1024 4096 10000; do
    env "$@" timeout 250 python bench.py --ctx $c --steps 8 --cpu-steps 0 --ttft-iters 5 --profile-iters 0 --ttft10k-iters 0 2>/dev/null | tail -1 |
      python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$label', 'tokens', d['config']['ctx'], 'ttft_ms', d['ttft_ms']['p50'])"
  done
}
{
run "all on              " X=1
run "GEMM128X256=0       " PEGAINFER_GEMM128X256=0
run "GEMM256=224         " PEGAINFER_GEMM256=224
run "PREFILL_DMA=0       " PEGAINFER_PREFILL_DMA=0
run "PREFILL_FUSE=0      " PEGAINFER_PREFILL_FUSE=0
run "all four off (r2)   " PEGAINFER_GEMM128X256=0 PEGAINFER_GEMM256=224 PEGAINFER_PREFILL_DMA=0 PEGAINFER_PREFILL_FUSE=0
run "all on              " X=1
} > gpurun_out/r3_prefill_features_ab.txt 2>&1
cat gpurun_out/r3_prefill_features_ab.txt
