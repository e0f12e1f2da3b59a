#!/bin/bash
# Run on the GPU box: rocprofv3 kernel trace of one bench.py invocation -> gpurun_out/<name>_kernel_stats.csv
# usage: tools/gpu_kt.sh <name> <bench.py args...>
repo=${GRAFT_REPO_ROOT:-/root/repo}
name=$1; shift
mkdir -p $repo/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_$name
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_$name -o run -- python $repo/bench.py "$@" > /tmp/prof_$name.log 2>&1
python $repo/tools/rocpd_stats.py "$(find /tmp/prof_$name -name '*.db' | head -1)" $repo/gpurun_out/${name}_kernel_stats.csv > /dev/null
grep '^{' /tmp/prof_$name.log | tail -1 > $repo/gpurun_out/${name}_bench_under_rocprof.json
head -16 $repo/gpurun_out/${name}_kernel_stats.csv | cut -c1-150
