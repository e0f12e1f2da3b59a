#!/bin/bash
repo=${GRAFT_REPO_ROOT:-/root/repo}
cd $repo
timeout 600 python -m pytest tests/test_gpu_real_dims.py -m gpu -x -q -k "engine or mode" > gpurun_out/pytest_eng.log 2>&1; tail -2 gpurun_out/pytest_eng.log
for m in 2 1 2; do
  timeout 300 python bench.py --decode-mode $m --steps 96 --cpu-steps 0 --ttft-iters 1 --profile-iters 0 --ttft10k-iters 0 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('mode $m tok/s', d['value'], 'device_ms', d['tpot_ms']['device_p50'])"
done
timeout 300 python tools/engine_probe.py --trace --steps 8 2>&1 | grep -v amdgpu.ids | tail -30
