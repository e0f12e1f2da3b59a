#!/bin/bash
# Round 5, third GPU call: (1) the 8B rerun diagnostic, (2) chained greedy decode - bit-identity tests, then the same-box
# A/B chain 32 / 0 - (3) the Qwen3.5-4B and Qwen3-8B bench lines with their new parity legs.
repo=${GRAFT_REPO_ROOT:-/root/repo}
cd $repo
mkdir -p gpurun_out
timeout 300 python tools/diag_8b.py > gpurun_out/r5_diag_8b.txt 2>&1
timeout 900 python -m pytest tests/test_gpu_model.py tests/test_gpu_fused.py -m gpu -q > gpurun_out/r5_chain_tests.log 2>&1
echo "pytest rc $?" >> gpurun_out/r5_chain_tests.log
: > gpurun_out/r5_chain_ab.txt
for x in 32 0 32 0 32 0; do
  timeout 300 python bench.py --chain $x --steps 192 --cpu-steps 0 --ttft-iters 1 --profile-iters 0 --ttft10k-iters 0 --sweep-steps 0 2>/dev/null | tail -1 | \
    python -c "import sys,json;d=json.loads(sys.stdin.read());print('--chain $x tok/s',d['value'],'ms_per_step',d['ms_per_step'],'device_ms',d['tpot_ms']['device_p50'],d['host_loop'])" >> gpurun_out/r5_chain_ab.txt
done
for b in 4 16; do for x in 32 0; do
  timeout 300 python bench.py --batch $b --chain $x --steps 96 --cpu-steps 0 --ttft-iters 1 --profile-iters 0 --ttft10k-iters 0 --sweep-steps 0 2>/dev/null | tail -1 | \
    python -c "import sys,json;d=json.loads(sys.stdin.read());print('bs $b --chain $x tok/s',d['value'],'ms_per_step',d['ms_per_step'],'device_ms',d['tpot_ms']['device_p50'])" >> gpurun_out/r5_chain_ab.txt
done; done
timeout 900 python bench.py --model qwen3.5-4b --steps 100 > gpurun_out/r5_qwen35_4b_bench.json 2> gpurun_out/r5_qwen35_4b_bench.err
timeout 1200 python bench.py --model qwen3-8b --steps 100 --sweep-steps 0 > gpurun_out/r5_qwen3_8b_greedy.json 2> gpurun_out/r5_qwen3_8b_greedy.err
timeout 300 python bench.py --model qwen3-8b --sampling topk_topp --steps 100 --cpu-steps 0 --sweep-steps 0 --profile-iters 0 --ttft10k-iters 0 > gpurun_out/r5_qwen3_8b_topk_topp.json 2>/dev/null
cat gpurun_out/r5_diag_8b.txt | grep -v amdgpu
grep -E "passed|failed|rc " gpurun_out/r5_chain_tests.log | tail -3
cat gpurun_out/r5_chain_ab.txt
tail -c 400 gpurun_out/r5_qwen35_4b_bench.json; echo; tail -c 300 gpurun_out/r5_qwen3_8b_greedy.json
