#!/bin/bash
# full gpu tests + TTFT sweep + decode kernel trace (bs 1) + bs 16 bench
repo=${GRAFT_REPO_ROOT:-/root/repo}
tag=${1:-r2c}
out=$repo/gpurun_out/$tag
mkdir -p $out
cd $repo
timeout 1200 python -m pytest tests -m gpu -x -q > $out/pytest.log 2>&1; echo "pytest rc $?" >> $out/pytest.log
tail -4 $out/pytest.log
for c in 1024 2048 4096 10000; do
  timeout 300 python bench.py --ctx $c --steps 32 --cpu-steps 0 --ttft-iters 3 2>/dev/null | tail -1 > $out/bench_ctx$c.json
  python -c "import json;d=json.load(open('$out/bench_ctx$c.json'));print('ctx',$c,d['value'],d['ms_per_step'],d.get('ttft_ms'))"
done
PEGAINFER_PREFILL_GROUP=1 timeout 300 python bench.py --ctx 4096 --steps 4 --cpu-steps 0 --ttft-iters 3 2>/dev/null | tail -1 | python -c "import sys,json;d=json.loads(sys.stdin.read());print('nogroup ctx4096',d.get('ttft_ms'))"
PEGAINFER_PREFILL_GROUP=2 timeout 300 python bench.py --ctx 1024 --steps 4 --cpu-steps 0 --ttft-iters 3 2>/dev/null | tail -1 | python -c "import sys,json;d=json.loads(sys.stdin.read());print('group2 ctx1024',d.get('ttft_ms'))"
for b in 16 32; do
  timeout 200 python bench.py --batch $b --steps 48 --cpu-steps 0 --ttft-iters 1 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('bs', d['config']['batch_per_gpu'], 'tok/s', d['value'], 'device_ms', d['tpot_ms']['device_p50'])"
done
bash tools/gpu_kt.sh ${tag}_decode --steps 64 --cpu-steps 0 --ttft-iters 2 | head -14
cp gpurun_out/${tag}_decode_kernel_stats.csv $out/ 2>/dev/null
