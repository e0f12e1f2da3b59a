#!/bin/bash
# One gpurun call (round 5): the three flush forms of skinny_resident_kernel (two barriers / one barrier / tickets), same
# box, cold weights, checksums of seeded products; then the op tests under the ticket form.
mkdir -p gpurun_out
out=gpurun_out/r5_skinny_flush_ab.txt
: > $out
run() { env "$@" timeout 150 python tools/bench_skinny.py ${SKINNY_TS:-4 8 16} 2>&1 | grep -v "amdgpu.ids" >> $out || echo "variant $* failed" >> $out; }
run PEGAINFER_SKINNY_FLUSH=0
run PEGAINFER_SKINNY_FLUSH=4
run PEGAINFER_SKINNY_FLUSH=1
run PEGAINFER_SKINNY_FLUSH=0
run PEGAINFER_SKINNY_FLUSH=4
grep layer4 $out
python - <<'PY'
import re, collections
d = collections.defaultdict(dict)
for l in open("gpurun_out/r5_skinny_flush_ab.txt"):
    m = re.match(r"\[(.*?)\] check (\S+) T=(\d+) sha=(\S+)", l)
    if m: d[(m.group(2), m.group(3))].setdefault(m.group(4), set()).add(m.group(1))
bad = {k: v for k, v in d.items() if len(v) > 1}
print("checksums equal across the flush forms:", not bad, bad if bad else "")
PY
for v in ${SKINNY_TEST_VARIANTS:-PEGAINFER_SKINNY_FLUSH=4}; do
  echo "== pytest under $v" | tee -a gpurun_out/r5_skinny_tests2.log
  env $v timeout 400 python -m pytest tests/test_gpu_fused.py tests/test_gpu_ops.py -x -q -m gpu 2>&1 | tail -4 | tee -a gpurun_out/r5_skinny_tests2.log
done
