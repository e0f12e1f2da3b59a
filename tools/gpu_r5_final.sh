#!/bin/bash
# Round 5, last call: the whole -m gpu suite with durations, the driver-style bench line, the batch sweep and the bs 16
# kernel stats on the final tree.
repo=${GRAFT_REPO_ROOT:-/root/repo}
cd $repo
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q --durations=15 > gpurun_out/r5_pytest_gpu_final.log 2>&1
echo "pytest rc $?" >> gpurun_out/r5_pytest_gpu_final.log
timeout 600 python bench.py --steps 20 --warmup 5 > gpurun_out/r5_bench_driver_style_final.json 2> gpurun_out/r5_bench_driver_style_final.err
echo "bench rc $?" >> gpurun_out/r5_bench_driver_style_final.err
for b in 2 4 8 16 32; do
  timeout 200 python bench.py --batch $b --steps 48 --cpu-steps 0 --ttft-iters 1 --profile-iters 0 --ttft10k-iters 0 --sweep-steps 0 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('bs', d['config']['batch_per_gpu'], 'tok/s', d['value'], 'device_ms', d['tpot_ms']['device_p50'])"
done > gpurun_out/r5_batch_sweep_final.txt
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_b16
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_b16 -o run -- python $repo/bench.py --batch 16 --steps 32 --cpu-steps 0 --ttft-iters 1 --profile-iters 0 --ttft10k-iters 0 --sweep-steps 0 > /tmp/prof_b16.log 2>&1
python $repo/tools/rocpd_stats.py "$(find /tmp/prof_b16 -name '*.db' | head -1)" $repo/gpurun_out/r5_batch16_final_kernel_stats.csv > /dev/null
cd $repo
grep -E "passed|failed|error|rc " gpurun_out/r5_pytest_gpu_final.log | tail -6
cat gpurun_out/r5_batch_sweep_final.txt
head -8 gpurun_out/r5_batch16_final_kernel_stats.csv
tail -c 600 gpurun_out/r5_bench_driver_style_final.json
