#!/bin/bash
repo=${GRAFT_REPO_ROOT:-/root/repo}
cd $repo
timeout 900 python -m pytest tests/test_gpu_fused.py tests/test_gpu_real_dims.py tests/test_gpu_model.py tests/test_gpu_ops.py -m gpu -x -q 2>&1 | tail -3
for x in 0 1 0 1; do
  PEGAINFER_GEMV_DYN=$x timeout 200 python bench.py --steps 96 --cpu-steps 0 --ttft-iters 1 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('dyn $x bs', d['config']['batch_per_gpu'], 'tok/s', d['value'], 'device_ms', d['tpot_ms']['device_p50'], d['gemv_sites'])"
done
PEGAINFER_GEMV_DYN=0 PEGAINFER_ATTN_WAVES=4 timeout 200 python bench.py --steps 96 --cpu-steps 0 --ttft-iters 1 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('attn waves 4: tok/s', d['value'], 'device_ms', d['tpot_ms']['device_p50'])"
timeout 300 python tools/gemv_probe.py --sites 1 5 3 7 2>&1 | grep -v amdgpu.ids
