#!/bin/bash
# Run on the GPU box (via gpurun): kernel-trace a short bench run and write the per-kernel table.
#   tools/gpu_profile.sh <out.csv> [bench args...]
out=$1; shift
repo=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_kt
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_kt -o run -- python $repo/bench.py "$@" > /tmp/prof_bench.log 2>&1
cd $repo
db=$(find /tmp/prof_kt -name '*.db' | head -1)
python tools/rocpd_stats.py "$db" "$out" | head -24
grep "^{" /tmp/prof_bench.log | tail -1 > "${out%.csv}_bench.json"
