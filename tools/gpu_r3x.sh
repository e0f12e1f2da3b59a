#!/bin/bash
# prefill attention A/B: register-staged vs LDS-DMA staging
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
{
for T in 10000 1024; do
  for v in 0 1; do
    PEGAINFER_PREFILL_DMA=$v timeout 120 python tools/bench_prefill_attn.py $T 2>&1 | grep -v amdgpu.ids
  done
done
} > gpurun_out/r3_prefill_attn_ab.txt 2>&1
cat gpurun_out/r3_prefill_attn_ab.txt
