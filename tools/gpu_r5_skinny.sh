#!/bin/bash
# One gpurun call (round 5): batched-decode GEMM variants at <= 16 token columns, same box, cold weights
# (tools/bench_skinny.py): the two-barrier flush of the resident kernel against the one-barrier form, a no-barrier timing
# probe, and the row-wave kernel; then the op tests under the variants that change a route.
mkdir -p gpurun_out
out=gpurun_out/r5_skinny_variants.txt
: > $out
run() { env "$@" timeout 150 python tools/bench_skinny.py ${SKINNY_TS:-4 16} 2>&1 | grep -v "amdgpu.ids" >> $out || echo "variant $* failed" >> $out; }
run X=0
run PEGAINFER_SKINNY_FLUSH=1
run PEGAINFER_SKINNY_FLUSH=2
run PEGAINFER_SKINNY_ROWWAVE=1
run PEGAINFER_SKINNY_ROWWAVE=1 PEGAINFER_SKINNY_ROWWAVE_RPB=8
cat $out
for v in ${SKINNY_TEST_VARIANTS:-PEGAINFER_SKINNY_ROWWAVE=1}; do
  echo "== pytest under $v" | tee -a gpurun_out/r5_skinny_tests.log
  env $v timeout 400 python -m pytest tests/test_gpu_fused.py tests/test_gpu_ops.py -x -q -m gpu 2>&1 | tail -4 | tee -a gpurun_out/r5_skinny_tests.log
done
