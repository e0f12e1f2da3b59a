#!/bin/bash
# Round 5: long-context decode, fused attention + o_proj plan: KV chunks rounded to whole rounds of the 8-wave workgroup
# (PEGAINFER_SPLIT_CHUNK_ALIGN 16 / 128) and 18 vs 20 chunks (PEGAINFER_OPROJ_CHUNKS), same box.
repo=${GRAFT_REPO_ROOT:-/root/repo}
cd $repo
mkdir -p gpurun_out
: > gpurun_out/r5_long_ctx_chunks_ab.txt
for c in 10000 4096 2048; do
  for cfg in "18 16" "18 128" "20 128" "20 16" "18 16" "20 128"; do
    set -- $cfg
    env PEGAINFER_OPROJ_CHUNKS=$1 PEGAINFER_SPLIT_CHUNK_ALIGN=$2 timeout 300 python bench.py --ctx $c --steps 64 --cpu-steps 0 --ttft-iters 1 --profile-iters 0 --ttft10k-iters 0 --sweep-steps 0 2>/dev/null | tail -1 | \
      python -c "import sys,json;d=json.loads(sys.stdin.read());print('ctx $c chunks $1 align $2 tok/s',d['value'],'device_ms',d['tpot_ms']['device_p50'])" >> gpurun_out/r5_long_ctx_chunks_ab.txt
  done
done
timeout 600 python -m pytest tests/test_gpu_fused.py tests/test_gpu_model.py "tests/test_gpu_real_dims.py::test_real_dims_bs1_prefill_1024_then_decode" "tests/test_gpu_real_dims.py::test_real_dims_two_requests_fused_attention_oproj" "tests/test_gpu_real_dims.py::test_real_dims_fused_path_bit_identical_to_reference_sequence" -m gpu -q > gpurun_out/r5_chunks_tests.log 2>&1
echo "pytest rc $?" >> gpurun_out/r5_chunks_tests.log
cat gpurun_out/r5_long_ctx_chunks_ab.txt
grep -E "passed|failed|rc " gpurun_out/r5_chunks_tests.log | tail -3
