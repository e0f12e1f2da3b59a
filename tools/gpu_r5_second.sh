#!/bin/bash
# Round 5, second GPU call: the Qwen3-8B x 36 diagnostic in a FRESH process (which of run A / its replays is off), the
# bit-identity tests the per-head-group hand-off of attn_oproj_kernel must keep, then the same-box A/B of that hand-off
# (PEGAINFER_OPROJ_GROUPWAIT 1 / 0 alternating) with the in-kernel phase trace of both forms.
repo=${GRAFT_REPO_ROOT:-/root/repo}
cd $repo
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_full_depth_8b.py -m gpu -q --durations=5 > gpurun_out/r5_8b_diag.log 2>&1
echo "pytest rc $?" >> gpurun_out/r5_8b_diag.log
timeout 900 python -m pytest tests/test_gpu_fused.py tests/test_gpu_model.py tests/test_gpu_real_dims.py -m gpu -q -x > gpurun_out/r5_groupwait_tests.log 2>&1
echo "pytest rc $?" >> gpurun_out/r5_groupwait_tests.log
: > gpurun_out/r5_oproj_groupwait_ab.txt
for x in 1 0 1 0 1 0; do
  env PEGAINFER_OPROJ_GROUPWAIT=$x timeout 300 python bench.py --steps 200 --cpu-steps 0 --ttft-iters 1 --profile-iters 0 --ttft10k-iters 0 --sweep-steps 0 2>/dev/null | tail -1 | \
    python -c "import sys,json;d=json.loads(sys.stdin.read());print('PEGAINFER_OPROJ_GROUPWAIT=$x tok/s',d['value'],'wall_p50',d['tpot_ms']['p50'],'device_ms',d['tpot_ms']['device_p50'])" >> gpurun_out/r5_oproj_groupwait_ab.txt
done
for x in 1 0; do
  echo "== PEGAINFER_OPROJ_GROUPWAIT=$x" >> gpurun_out/r5_oproj_groupwait_ab.txt
  env PEGAINFER_OPROJ_GROUPWAIT=$x timeout 120 python tools/attn_probe.py --ctx 1024 --layers 8 >> gpurun_out/r5_oproj_groupwait_ab.txt 2>&1
done
for x in 1 0; do
  env PEGAINFER_OPROJ_GROUPWAIT=$x timeout 300 python bench.py --ctx 4096 --steps 100 --cpu-steps 0 --ttft-iters 1 --profile-iters 0 --ttft10k-iters 0 --sweep-steps 0 2>/dev/null | tail -1 | \
    python -c "import sys,json;d=json.loads(sys.stdin.read());print('ctx 4096 PEGAINFER_OPROJ_GROUPWAIT=$x tok/s',d['value'],'device_ms',d['tpot_ms']['device_p50'])" >> gpurun_out/r5_oproj_groupwait_ab.txt
done
grep -E "passed|failed|rc " gpurun_out/r5_8b_diag.log gpurun_out/r5_groupwait_tests.log | tail -6
cat gpurun_out/r5_oproj_groupwait_ab.txt | grep -v "^\s*$" | tail -40
