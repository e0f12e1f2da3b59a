#!/bin/bash
repo=${GRAFT_REPO_ROOT:-/root/repo}
cd $repo
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_real_dims.py tests/test_gpu_qwen35.py tests/test_gpu_model.py -m gpu -x -q -k "prefill or model or real or unified" 2>&1 | tail -2
for x in 1 0 1 0; do
  for c in 1024 4096 10000; do
    PEGAINFER_PREFILL_XCD_HEADS=$x timeout 300 python bench.py --ctx $c --steps 4 --cpu-steps 0 --ttft-iters 3 2>/dev/null | tail -1 | python -c "import sys,json;d=json.loads(sys.stdin.read());print('xcd_heads $x ctx',$c,d.get('ttft_ms'))"
  done
done
