#!/bin/bash
repo=${GRAFT_REPO_ROOT:-/root/repo}
tag=${1:-r2e}
out=$repo/gpurun_out/$tag
mkdir -p $out
cd $repo
timeout 1200 python -m pytest tests -m gpu -x -q > $out/pytest.log 2>&1; echo "pytest rc $?" >> $out/pytest.log
tail -4 $out/pytest.log
timeout 300 python tools/attn_probe.py --ctx 1024 2>&1 | tail -9
timeout 300 python tools/attn_probe.py --ctx 1024 --batch 16 2>&1 | tail -9
timeout 300 python bench.py --steps 128 --cpu-steps 0 --ttft-iters 2 2>/dev/null | tail -1 > $out/bench.json
python -c "import json;d=json.load(open('$out/bench.json'));print('bs1',d['value'],d['ms_per_step'],d['tpot_ms'])"
for b in 8 16 32; do
  timeout 200 python bench.py --batch $b --steps 48 --cpu-steps 0 --ttft-iters 1 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('bs', d['config']['batch_per_gpu'], 'tok/s', d['value'], 'device_ms', d['tpot_ms']['device_p50'])"
done
