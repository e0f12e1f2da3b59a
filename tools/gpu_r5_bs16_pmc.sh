#!/bin/bash
# Round 5: HBM bytes fetched per dispatch (FETCH_SIZE, one counter, --kernel-trace only) of the bs 16 decode kernels.
repo=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $repo/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pmc_b16
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d /tmp/pmc_b16 -o run -- python $repo/bench.py --batch 16 --steps 8 --cpu-steps 0 --ttft-iters 1 --profile-iters 0 --ttft10k-iters 0 --sweep-steps 0 > /tmp/pmc_b16.log 2>&1
python $repo/tools/rocpd_pmc.py "$(find /tmp/pmc_b16 -name '*.db' | head -1)" $repo/gpurun_out/r5_batch16_pmc_FETCH_SIZE.csv
head -12 $repo/gpurun_out/r5_batch16_pmc_FETCH_SIZE.csv
