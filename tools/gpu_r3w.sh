#!/bin/bash
# A/B of the 128 x 256-tile prefill GEMM (PEGAINFER_GEMM128X256) on the projection shapes, cold weights
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
for T in 1024 512 2048; do
  for v in 0 1; do
    PEGAINFER_GEMM128X256=$v timeout 120 python tools/bench_prefill_gemm.py $T 12 2>&1 | grep -v amdgpu.ids | sed "s/^/g128=$v /"
  done
done > gpurun_out/r3_gemm128x256_ab.txt 2>&1
cat gpurun_out/r3_gemm128x256_ab.txt
