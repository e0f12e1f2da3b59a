#!/bin/bash
# One gpurun call (round 5): the K-split dot2 GEMV (down_proj at bs 1; every GEMV of a hidden-4096 model) with the
# barrier-free ticket reduction (PEGAINFER_GEMV_TICKET=1) against the barrier form, same box, alternating; then the
# op / model tests under the ticket form (bit-identity with the reference sequence included).
repo=${GRAFT_REPO_ROOT:-/root/repo}
cd $repo
mkdir -p gpurun_out
ab=gpurun_out/r5_gemv_ticket_ab.txt
: > $ab
one() {  # model, knob
  env PEGAINFER_GEMV_TICKET=$2 timeout 300 python bench.py --model $1 --steps 96 --cpu-steps 0 --ttft-iters 1 --profile-iters 0 --ttft10k-iters 0 --sweep-steps 0 2>/dev/null | tail -1 | \
    python -c "import sys,json;d=json.loads(sys.stdin.read());print('$1 ticket $2 tok/s',d['value'],'device_ms',d['tpot_ms']['device_p50'],'sites',{k:v.get('us') for k,v in d.get('gemv_sites',{}).items()} if isinstance(d.get('gemv_sites'),dict) else '')" >> $ab
}
for rep in 1 2; do one qwen3-4b 0; one qwen3-4b 1; done
one qwen3-8b 0; one qwen3-8b 1
cat $ab
PEGAINFER_GEMV_TICKET=1 timeout 600 python -m pytest tests/test_gpu_fused.py tests/test_gpu_ops.py tests/test_gpu_model.py "tests/test_gpu_real_dims.py::test_real_dims_bs1_prefill_1024_then_decode" "tests/test_gpu_real_dims.py::test_real_dims_fused_path_bit_identical_to_reference_sequence" -m gpu -q -x > gpurun_out/r5_gemv_ticket_tests.log 2>&1
echo "pytest rc $?" >> gpurun_out/r5_gemv_ticket_tests.log
grep -E "passed|failed|rc " gpurun_out/r5_gemv_ticket_tests.log | tail -3
