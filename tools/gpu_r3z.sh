#!/bin/bash
# Two SQ PMC passes over the prefill-attention microbench (10 k tokens): where do the waves of the LDS-DMA kernel wait?
repo=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
mkdir -p $repo/gpurun_out
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_VALU_MFMA_BUSY_CYCLES" \
           "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INST_CYCLES_SALU"; do
  i=$((i+1))
  rm -rf /tmp/pmc_pf$i
  timeout 300 rocprofv3 --pmc $set --kernel-trace -d /tmp/pmc_pf$i -o run -- python $repo/tools/bench_prefill_attn.py 10000 > /tmp/pmc_pf$i.log 2>&1
  tail -2 /tmp/pmc_pf$i.log | cut -c1-200
  python $repo/tools/rocpd_pmc.py "$(find /tmp/pmc_pf$i -name '*.db' | head -1)" $repo/gpurun_out/r3_prefill_attn_pmc$i.csv
done
grep -h batch_prefill $repo/gpurun_out/r3_prefill_attn_pmc1.csv $repo/gpurun_out/r3_prefill_attn_pmc2.csv | cut -c1-200
