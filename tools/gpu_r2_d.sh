#!/bin/bash
repo=${GRAFT_REPO_ROOT:-/root/repo}
tag=${1:-r2d}
out=$repo/gpurun_out/$tag
mkdir -p $out
cd $repo
timeout 1200 python -m pytest tests -m gpu -x -q > $out/pytest.log 2>&1; echo "pytest rc $?" >> $out/pytest.log
tail -4 $out/pytest.log
timeout 300 python tools/attn_probe.py --ctx 1024 2>&1 | tail -12
timeout 300 python tools/attn_probe.py --ctx 1024 --batch 16 2>&1 | tail -12
