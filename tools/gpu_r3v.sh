#!/bin/bash
repo=${GRAFT_REPO_ROOT:-/root/repo}
cd $repo
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_fused.py tests/test_gpu_model.py tests/test_gpu_real_dims.py -m gpu -q -x --tb=short > gpurun_out/pytest_r3v.log 2>&1
tail -6 gpurun_out/pytest_r3v.log
for i in 1 2; do
  for f in 0 1; do
    for c in 1024 128 4096; do
      PEGAINFER_PREFILL_FUSE=$f timeout 200 python bench.py --ctx $c --steps 4 --cpu-steps 0 --ttft-iters 10 --profile-iters 0 --ttft10k-iters 0 2>/dev/null | tail -1 | python -c "import sys,json;d=json.loads(sys.stdin.read());print('prefill_fuse=$f ctx',$c,d['ttft_ms'])"
    done
  done
done 2>&1 | tee gpurun_out/r3v_prefill_fuse_ab.txt
