#!/bin/bash
# prefill attention check: parity tests touching prefill + TTFT at 1024/4096/10000 + kernel trace at 10000
repo=${GRAFT_REPO_ROOT:-/root/repo}
tag=${1:-r2b}
out=$repo/gpurun_out/$tag
mkdir -p $out
cd $repo
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_real_dims.py tests/test_gpu_qwen35.py tests/test_gpu_model.py -m gpu -x -q -k "prefill or model or real or unified" > $out/pytest.log 2>&1; echo "pytest rc $?" >> $out/pytest.log
tail -4 $out/pytest.log
for c in 1024 4096 10000; do
  timeout 300 python bench.py --ctx $c --steps 8 --cpu-steps 0 --ttft-iters 3 2>/dev/null | tail -1 > $out/bench_ctx$c.json
  python -c "import json;d=json.load(open('$out/bench_ctx$c.json'));print('ctx',$c,d['value'],d['ms_per_step'],d.get('ttft_ms'))"
done
bash tools/gpu_kt.sh ${tag}_ctx10000 --ctx 10000 --steps 4 --cpu-steps 0 --ttft-iters 2 | head -12
cp gpurun_out/${tag}_ctx10000_kernel_stats.csv $out/ 2>/dev/null
