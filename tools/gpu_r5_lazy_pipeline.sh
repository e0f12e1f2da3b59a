#!/bin/bash
# Round 5: lazy tickets (the default) against waiting tickets (PEGAINFER_SKINNY_FLUSH=4) in the pipeline at bs 16, same box,
# alternating; then the model tests that run 3..16-column decode steps (tiny golden model + real-width two-layer model).
cd ${GRAFT_REPO_ROOT:-/root/repo}; mkdir -p gpurun_out; ab=gpurun_out/r5_batch_lazy_ab.txt; : > $ab
for rep in 1 2; do for v in PEGAINFER_SKINNY_FLUSH=4 X=0; do
  env $v timeout 200 python bench.py --batch ${1:-16} --steps 48 --cpu-steps 0 --ttft-iters 1 --profile-iters 0 --ttft10k-iters 0 --sweep-steps 0 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v bs', d['config']['batch_per_gpu'], 'tok/s', d['value'], 'device_ms', d['tpot_ms']['device_p50'])" >> $ab
done; done; cat $ab
timeout 400 python -m pytest tests/test_gpu_fused.py tests/test_gpu_model.py "tests/test_gpu_real_dims.py::test_real_dims_batched_prefill_and_decode" "tests/test_gpu_real_dims.py::test_gemm_lm_head_qwen3_8b_and_qwen35" -m gpu -q -x 2>&1 | tail -3
