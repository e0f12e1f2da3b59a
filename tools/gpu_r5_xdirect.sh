#!/bin/bash
# One gpurun call (round 5): down_proj at 7..16 columns (tiled skinny kernel), x by LDS-DMA tiles + barriers against x
# fragments straight into registers (PEGAINFER_SKINNY_XDIRECT=1); checksums must agree.
mkdir -p gpurun_out
out=gpurun_out/r5_skinny_xdirect_ab.txt
: > $out
run() { env "$@" timeout 150 python tools/bench_skinny.py ${SKINNY_TS:-8 16} 2>&1 | grep -v "amdgpu.ids" >> $out || echo "variant $* failed" >> $out; }
run PEGAINFER_SKINNY_XDIRECT=0
run PEGAINFER_SKINNY_XDIRECT=1
run PEGAINFER_SKINNY_XDIRECT=0
run PEGAINFER_SKINNY_XDIRECT=1
grep layer4 $out | sed -e 's/| lm_head.*//'
python - <<'PY'
import re, collections
d = collections.defaultdict(dict)
for l in open("gpurun_out/r5_skinny_xdirect_ab.txt"):
    m = re.match(r"\[(.*?)\] check (\S+) T=(\d+) sha=(\S+)", l)
    if m: d[(m.group(2), m.group(3))].setdefault(m.group(4), set()).add(m.group(1))
bad = {k: v for k, v in d.items() if len(v) > 1}
print("checksums equal across the forms:", not bad, bad if bad else "")
PY
