#!/bin/bash
# One gpurun call (round 5): ticket flush (4) against the lazy ticket form (5) and the two-barrier form (0) at the call
# sites, then the flush-form bit-identity test + the fused op tests under the lazy form.
mkdir -p gpurun_out
out=gpurun_out/r5_skinny_flush_ab3.txt
: > $out
run() { env "$@" timeout 150 python tools/bench_skinny.py ${SKINNY_TS:-4 16} 2>&1 | grep -v "amdgpu.ids" >> $out || echo "variant $* failed" >> $out; }
run PEGAINFER_SKINNY_FLUSH=5
run PEGAINFER_SKINNY_FLUSH=4
run PEGAINFER_SKINNY_FLUSH=0
run PEGAINFER_SKINNY_FLUSH=5
run PEGAINFER_SKINNY_FLUSH=4
grep layer4 $out
python - <<'PY'
import re, collections
d = collections.defaultdict(dict)
for l in open("gpurun_out/r5_skinny_flush_ab3.txt"):
    m = re.match(r"\[(.*?)\] check (\S+) T=(\d+) sha=(\S+)", l)
    if m: d[(m.group(2), m.group(3))].setdefault(m.group(4), set()).add(m.group(1))
bad = {k: v for k, v in d.items() if len(v) > 1}
print("checksums equal across the flush forms:", not bad, bad if bad else "")
PY
PEGAINFER_SKINNY_FLUSH=5 timeout 300 python -m pytest tests/test_gpu_fused.py -x -q -m gpu 2>&1 | tail -3
