#!/bin/bash
# One gpurun call: skinny GEMM call-site timings (tools/bench_skinny.py) for each variant in SKINNY_VARIANTS
# (space-separated VAR=value settings, e.g. "X=0 PEGAINFER_SKINNY_RB=1"), same box, same weights.
mkdir -p gpurun_out
out=gpurun_out/skinny_probe.log
: > $out
for v in ${SKINNY_VARIANTS:-X=0}; do
  env $v timeout 120 python tools/bench_skinny.py ${SKINNY_TS:-8 16} >> $out 2>&1 || echo "variant $v failed" >> $out
done
grep -v "amdgpu.ids" $out
