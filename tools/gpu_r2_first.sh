#!/bin/bash
# Round-2 first GPU pass: gpu tests, bench at decode modes 1/2, prefill TTFT at 1024/4096/10000.
repo=${GRAFT_REPO_ROOT:-/root/repo}
out=$repo/gpurun_out/r2a
mkdir -p $out
cd $repo
timeout 1500 python -m pytest tests -m gpu -x -q > $out/pytest.log 2>&1; echo "pytest rc $?" >> $out/pytest.log
tail -5 $out/pytest.log
for m in 1 2; do
  timeout 300 python bench.py --decode-mode $m --steps 128 --cpu-steps 0 --ttft-iters 3 2>$out/bench_m$m.err | tail -1 > $out/bench_m$m.json
  python -c "import json;d=json.load(open('$out/bench_m$m.json'));print('mode',$m,d['value'],d['ms_per_step'],d.get('ttft_ms'),d.get('step_roofline'))"
done
for c in 4096 10000; do
  timeout 300 python bench.py --ctx $c --steps 16 --cpu-steps 0 --ttft-iters 2 2>/dev/null | tail -1 > $out/bench_ctx$c.json
  python -c "import json;d=json.load(open('$out/bench_ctx$c.json'));print('ctx',$c,d['value'],d['ms_per_step'],d.get('ttft_ms'))"
done
