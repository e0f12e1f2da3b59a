#!/bin/bash
# Run on the GPU box: one PMC pass with the counters in $PMC over a bench.py invocation -> gpurun_out/<name>_pmc.csv
# usage: PMC="SQ_INSTS_VALU SQ_WAVE_CYCLES ..." tools/gpu_pmc_custom.sh <name> <bench args...>
name=${1:-custom}; shift
repo=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pmc_c
timeout 400 rocprofv3 --pmc $PMC --kernel-trace -d /tmp/pmc_c -o run -- python $repo/bench.py --steps 8 --cpu-steps 0 --ttft-iters 1 --profile-iters 0 --ttft10k-iters 0 "$@" > /tmp/pmc_c.log 2>&1
tail -2 /tmp/pmc_c.log | cut -c1-200
python $repo/tools/rocpd_pmc.py "$(find /tmp/pmc_c -name '*.db' | head -1)" $repo/gpurun_out/${name}_pmc.csv
grep "decode_attn\|skinny_resident_kernel<1, 1" $repo/gpurun_out/${name}_pmc.csv | cut -c1-140
