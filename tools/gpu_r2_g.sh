#!/bin/bash
repo=${GRAFT_REPO_ROOT:-/root/repo}
out=$repo/gpurun_out/r2g
mkdir -p $out
cd $repo
timeout 1200 python -m pytest tests -m gpu -x -q > $out/pytest.log 2>&1; echo "pytest rc $?" >> $out/pytest.log
tail -3 $out/pytest.log
sw() { # dir label
  for b in 1 2 16 32; do
    (cd $1 && timeout 200 python bench.py --batch $b --steps 64 --cpu-steps 0 --ttft-iters 1 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$2 bs', d['config']['batch_per_gpu'], 'tok/s', d['value'], 'device_ms', d['tpot_ms']['device_p50'])")
  done
}
sw . new
sw _ab base
timeout 300 python tools/gemv_probe.py --sites 6 1 5 3 2>&1 | grep -v amdgpu.ids
