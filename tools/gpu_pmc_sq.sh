#!/bin/bash
# Run on the GPU box: one SQ PMC pass (8 slots) over a short bench run -> gpurun_out/<name>_pmc_sq.csv
name=${1:-r1}; shift
repo=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pmc_sq
timeout 400 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY \
  --kernel-trace -d /tmp/pmc_sq -o run -- python $repo/bench.py --steps 8 --cpu-steps 0 --ttft-iters 2 --profile-iters 0 --ttft10k-iters 0 "$@" > /tmp/pmc_sq.log 2>&1
tail -3 /tmp/pmc_sq.log | cut -c1-300
python $repo/tools/rocpd_pmc.py "$(find /tmp/pmc_sq -name '*.db' | head -1)" $repo/gpurun_out/${name}_pmc_sq.csv
head -60 $repo/gpurun_out/${name}_pmc_sq.csv
