#!/bin/bash
# round 3, call A: the new tests first (fail fast and loud), then the whole -m gpu suite, then the default bench line
repo=${GRAFT_REPO_ROOT:-/root/repo}
cd $repo
mkdir -p gpurun_out
t0=$(date +%s)
timeout 900 python -m pytest tests/test_gpu_comm.py tests/test_gpu_comm_multi.py tests/test_gpu_real_dims_cfg34.py tests/test_gpu_qwen35.py \
  "tests/test_gpu_real_dims.py" -m gpu -q --tb=short --durations=15 -k "ep_ or oneshot or two_gpus or cfg34 or qwen3_8b_shape or qwen35_real or hd256 or gdr_chunkwise or engine" \
  > gpurun_out/pytest_new.log 2>&1
echo "pytest rc $?" >> gpurun_out/pytest_new.log
grep -E "passed|failed|error|rc |^FAILED|^ERROR|assert" gpurun_out/pytest_new.log | tail -40
echo "new tests took $(( $(date +%s) - t0 )) s"
t0=$(date +%s)
timeout 1200 python -m pytest tests -m gpu -q --tb=short > gpurun_out/pytest_gpu.log 2>&1
echo "pytest rc $?" >> gpurun_out/pytest_gpu.log
grep -E "passed|failed|error|rc |^FAILED|^ERROR" gpurun_out/pytest_gpu.log | tail -20
echo "full suite took $(( $(date +%s) - t0 )) s"
t0=$(date +%s)
timeout 600 python bench.py --steps 20 > gpurun_out/r3a_bench_default.json 2> gpurun_out/r3a_bench_default.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/r3a_bench_default.json'))
print('value',d['value'],'tpot',d['tpot_ms'],'ttft',d['ttft_ms'],'heavy',d.get('decode_heavy'),'10k',d.get('ttft_ms_10000'))
print('roofline',d['roofline']['frac'],d['roofline']['avg_launch_us'],'step',d['step_roofline'],'sites',d.get('gemv_sites'))
PY
echo "bench took $(( $(date +%s) - t0 )) s"
