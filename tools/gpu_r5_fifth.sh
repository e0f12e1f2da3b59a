#!/bin/bash
# Round 5, fifth GPU call: (1) diag third pass (graph vs eager truth, magnitudes, knock-outs), (2) the re-run path test +
# chain tests on the token-slot alias, (3) chain A/B at bs 1 / 16 again (no D2D copy now).
repo=${GRAFT_REPO_ROOT:-/root/repo}
cd $repo
mkdir -p gpurun_out
timeout 900 python tools/diag_8b.py > gpurun_out/r5_diag_8b_3.txt 2>&1
timeout 900 python -m pytest tests/test_gpu_model.py "tests/test_gpu_real_dims.py::test_real_dims_fused_attention_oproj_expired_wait_reruns_the_step" "tests/test_gpu_real_dims.py::test_real_dims_two_requests_fused_attention_oproj" tests/test_gpu_logprobs.py -m gpu -q > gpurun_out/r5_alias_tests.log 2>&1
echo "pytest rc $?" >> gpurun_out/r5_alias_tests.log
: > gpurun_out/r5_chain_ab2.txt
for b in 1 16; do for x in 32 0 32 0; do
  timeout 300 python bench.py --batch $b --chain $x --steps 128 --cpu-steps 0 --ttft-iters 1 --profile-iters 0 --ttft10k-iters 0 --sweep-steps 0 2>/dev/null | tail -1 | \
    python -c "import sys,json;d=json.loads(sys.stdin.read());print('bs $b --chain $x tok/s',d['value'],'ms_per_step',d['ms_per_step'],'device_ms',d['tpot_ms']['device_p50'])" >> gpurun_out/r5_chain_ab2.txt
done; done
grep -v amdgpu gpurun_out/r5_diag_8b_3.txt
grep -E "passed|failed|rc |Error" gpurun_out/r5_alias_tests.log | tail -6
cat gpurun_out/r5_chain_ab2.txt
