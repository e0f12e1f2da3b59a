#!/bin/bash
# Round 5, first GPU call: the CU-subset ingest probe, the new full-depth parity modules (+ the GPU tests whose code paths the
# ADVICE fixes touched), then the default bench line.  Everything lands in gpurun_out/ (gpurun returns only a tail).
repo=${GRAFT_REPO_ROOT:-/root/repo}
cd $repo
mkdir -p gpurun_out
{ nproc; free -g | head -2; rocm-smi --showuse 2>/dev/null | head -8; } > gpurun_out/r5_host.txt 2>&1
timeout 120 ./tools/probes/ingest_subset_probe > gpurun_out/r5_ingest_subset.txt 2>&1
echo "probe rc $?" >> gpurun_out/r5_ingest_subset.txt
timeout 1500 python -m pytest tests/test_gpu_full_depth.py tests/test_gpu_full_depth_qwen35.py tests/test_gpu_full_depth_8b.py \
    tests/test_gpu_tp_one_gpu.py tests/test_gpu_logprobs.py "tests/test_gpu_qwen35_model.py::test_qwen35_scheduler_sampling_and_logprobs" \
    tests/test_gpu_comm_multi.py -m gpu -q --durations=20 > gpurun_out/r5_depth_tests.log 2>&1
echo "pytest rc $?" >> gpurun_out/r5_depth_tests.log
timeout 600 python bench.py > gpurun_out/r5_bench_first.json 2> gpurun_out/r5_bench_first.err
echo "bench rc $?" >> gpurun_out/r5_bench_first.err
grep -E "passed|failed|error|rc " gpurun_out/r5_depth_tests.log | tail -8
tail -3 gpurun_out/r5_ingest_subset.txt
tail -c 600 gpurun_out/r5_bench_first.json
