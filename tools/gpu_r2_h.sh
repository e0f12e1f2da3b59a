#!/bin/bash
repo=${GRAFT_REPO_ROOT:-/root/repo}
cd $repo
for x in 0 1 0 1; do
  PEGAINFER_GEMV_XWAIT=$x timeout 200 python bench.py --steps 96 --cpu-steps 0 --ttft-iters 1 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('xwait $x bs', d['config']['batch_per_gpu'], 'tok/s', d['value'], 'device_ms', d['tpot_ms']['device_p50'])"
done
PEGAINFER_GEMV_XWAIT=1 timeout 300 python tools/gemv_probe.py --sites 6 1 5 3 2>&1 | grep -v amdgpu.ids
