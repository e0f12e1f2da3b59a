#!/bin/bash
repo=${GRAFT_REPO_ROOT:-/root/repo}
cd $repo
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_fused.py tests/test_gpu_real_dims.py -m gpu -x -q -k "attention or decode or real or fused" > gpurun_out/pytest_attn.log 2>&1; tail -2 gpurun_out/pytest_attn.log
for x in 1 0 1 0; do
  PEGAINFER_ATTN_LIVE_PARTS=$x timeout 200 python bench.py --steps 96 --cpu-steps 0 --ttft-iters 1 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('live $x bs1 tok/s', d['value'], 'device_ms', d['tpot_ms']['device_p50'])"
done
for x in 1 0; do
  PEGAINFER_ATTN_LIVE_PARTS=$x timeout 200 python bench.py --batch 16 --steps 48 --cpu-steps 0 --ttft-iters 1 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('live $x bs16 tok/s', d['value'], 'device_ms', d['tpot_ms']['device_p50'])"
done
