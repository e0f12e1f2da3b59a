#!/bin/bash
# Round 5, fourth GPU call: (1) which state a sample() call damages on the 36-layer 8B engine, (2) the two-request form of the
# fused attention + o_proj launch: bit-identity tests, then its same-box A/B at bs 2 (PEGAINFER_OPROJ_MAX_BATCH 2 / 1).
repo=${GRAFT_REPO_ROOT:-/root/repo}
cd $repo
mkdir -p gpurun_out
timeout 600 python tools/diag_8b.py > gpurun_out/r5_diag_8b_2.txt 2>&1
timeout 1200 python -m pytest tests/test_gpu_real_dims.py tests/test_gpu_fused.py tests/test_gpu_model.py -m gpu -q -k "not gemm_qwen3 and not lm_head" > gpurun_out/r5_bs2_tests.log 2>&1
echo "pytest rc $?" >> gpurun_out/r5_bs2_tests.log
: > gpurun_out/r5_bs2_oproj_ab.txt
for x in 2 1 2 1; do
  env PEGAINFER_OPROJ_MAX_BATCH=$x timeout 300 python bench.py --batch 2 --steps 128 --cpu-steps 0 --ttft-iters 1 --profile-iters 0 --ttft10k-iters 0 --sweep-steps 0 2>/dev/null | tail -1 | \
    python -c "import sys,json;d=json.loads(sys.stdin.read());print('bs 2 PEGAINFER_OPROJ_MAX_BATCH=$x tok/s',d['value'],'ms_per_step',d['ms_per_step'],'device_ms',d['tpot_ms']['device_p50'])" >> gpurun_out/r5_bs2_oproj_ab.txt
done
for c in 512 2048; do for x in 2 1; do
  env PEGAINFER_OPROJ_MAX_BATCH=$x timeout 300 python bench.py --batch 2 --ctx $c --steps 96 --cpu-steps 0 --ttft-iters 1 --profile-iters 0 --ttft10k-iters 0 --sweep-steps 0 2>/dev/null | tail -1 | \
    python -c "import sys,json;d=json.loads(sys.stdin.read());print('bs 2 ctx $c PEGAINFER_OPROJ_MAX_BATCH=$x tok/s',d['value'],'device_ms',d['tpot_ms']['device_p50'])" >> gpurun_out/r5_bs2_oproj_ab.txt
done; done
grep -v amdgpu gpurun_out/r5_diag_8b_2.txt
grep -E "passed|failed|rc |Error" gpurun_out/r5_bs2_tests.log | tail -6
cat gpurun_out/r5_bs2_oproj_ab.txt
