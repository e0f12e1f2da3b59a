#!/bin/bash
# One gpurun call (round 5): ticket flush with a designated reducer (4) against the two-barrier form (0) at the call sites (the
# run archived as profiles/r5_skinny_flush_ab2.txt also had a last-arriver ticket form, 12, measured slower and removed since); then the batch sweep points with the two-barrier form against the default (auto) in the pipeline.
mkdir -p gpurun_out
out=gpurun_out/r5_skinny_flush_ab2.txt
: > $out
run() { env "$@" timeout 150 python tools/bench_skinny.py ${SKINNY_TS:-4 16} 2>&1 | grep -v "amdgpu.ids" >> $out || echo "variant $* failed" >> $out; }
run PEGAINFER_SKINNY_FLUSH=4
run PEGAINFER_SKINNY_FLUSH=0
run PEGAINFER_SKINNY_FLUSH=4
run PEGAINFER_SKINNY_FLUSH=0
grep layer4 $out
python - <<'PY'
import re, collections
d = collections.defaultdict(dict)
for l in open("gpurun_out/r5_skinny_flush_ab2.txt"):
    m = re.match(r"\[(.*?)\] check (\S+) T=(\d+) sha=(\S+)", l)
    if m: d[(m.group(2), m.group(3))].setdefault(m.group(4), set()).add(m.group(1))
bad = {k: v for k, v in d.items() if len(v) > 1}
print("checksums equal across the flush forms:", not bad, bad if bad else "")
PY
ab=gpurun_out/r5_batch_flush_ab.txt
: > $ab
for rep in 1 2; do
  for b in ${SWEEP_BATCHES:-4 16}; do
    for v in PEGAINFER_SKINNY_FLUSH=0 X=0; do
      env $v timeout 200 python bench.py --batch $b --steps 48 --cpu-steps 0 --ttft-iters 1 --profile-iters 0 --ttft10k-iters 0 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v bs', d['config']['batch_per_gpu'], 'tok/s', d['value'], 'device_ms', d['tpot_ms']['device_p50'])" >> $ab
    done
  done
done
cat $ab
