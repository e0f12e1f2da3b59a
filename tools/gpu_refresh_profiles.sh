#!/bin/bash
# Run on the GPU box (via gpurun): regenerate the files under profiles/ for round $1 (default r2) into
# gpurun_out/profiles_new/.  part = a | b | all (two calls keep each under ~8 minutes of box time).
R=${1:-r3}; part=${2:-all}
repo=${GRAFT_REPO_ROOT:-/root/repo}
out=$repo/gpurun_out/profiles_new
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
kt() {  # name, bench args...
  local name=$1; shift
  rm -rf /tmp/prof_$name
  timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_$name -o run -- python $repo/bench.py "$@" > /tmp/prof_$name.log 2>&1
  python $repo/tools/rocpd_stats.py "$(find /tmp/prof_$name -name '*.db' | head -1)" $out/${name}_kernel_stats.csv > /dev/null
  grep '^{' /tmp/prof_$name.log | tail -1 > $out/${name}_bench_under_rocprof.json
}
pmc() {  # out name, counters, bench args...
  local name=$1 ctr=$2; shift 2
  rm -rf /tmp/pmc_$name
  timeout 300 rocprofv3 --pmc $ctr --kernel-trace -d /tmp/pmc_$name -o run -- python $repo/bench.py --cpu-steps 0 "$@" > /tmp/pmc_$name.log 2>&1
  python $repo/tools/rocpd_pmc.py "$(find /tmp/pmc_$name -name '*.db' | head -1)" $out/${name}.csv
}
SQ="SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"
if [ $part = a ] || [ $part = all ]; then
  kt ${R}_fused_decode_mode1 --steps 64 --cpu-steps 0 --ttft-iters 5 --profile-iters 0 --ttft10k-iters 0
  kt ${R}_baseline_decode_mode0 --steps 64 --cpu-steps 0 --decode-mode 0 --ttft-iters 1 --profile-iters 0 --ttft10k-iters 0
  kt ${R}_ctx10000 --ctx 10000 --steps 8 --cpu-steps 0 --ttft-iters 2 --profile-iters 0 --ttft10k-iters 0
  pmc ${R}_fused_pmc_FETCH_SIZE FETCH_SIZE --steps 8 --ttft-iters 1 --profile-iters 0 --ttft10k-iters 0
  pmc ${R}_fused_pmc_WRITE_SIZE WRITE_SIZE --steps 8 --ttft-iters 1 --profile-iters 0 --ttft10k-iters 0
  pmc ${R}_fused_pmc_sq "$SQ" --steps 8 --ttft-iters 2 --profile-iters 0 --ttft10k-iters 0
  pmc ${R}_ctx10000_pmc_sq "$SQ" --ctx 10000 --steps 4 --ttft-iters 1 --profile-iters 0 --ttft10k-iters 0
  cd $repo
  timeout 400 python bench.py > $out/${R}_bench_default_run.json 2>/dev/null
  timeout 200 python tools/attn_probe.py --ctx 1024 2>&1 | grep -v amdgpu.ids > $out/${R}_attn_phase_trace.txt
  timeout 200 python tools/attn_probe.py --ctx 1024 --batch 16 2>&1 | grep -v amdgpu.ids >> $out/${R}_attn_phase_trace.txt
  timeout 200 python tools/gemv_probe.py --sites 6 1 5 3 7 2>&1 | grep -v amdgpu.ids > $out/${R}_gemv_phase_trace.txt
  timeout 200 python tools/attn_probe.py --ctx 4096 2>&1 | grep -v amdgpu.ids >> $out/${R}_attn_phase_trace.txt
  timeout 200 python tools/attn_probe.py --ctx 10000 2>&1 | grep -v amdgpu.ids >> $out/${R}_attn_phase_trace.txt
fi
if [ $part = b ] || [ $part = all ]; then
  cd $repo
  for b in 2 4 8 16 32 64; do
    timeout 200 python bench.py --batch $b --steps 48 --cpu-steps 0 --ttft-iters 1 --profile-iters 0 --ttft10k-iters 0 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('bs', d['config']['batch_per_gpu'], 'tok/s', d['value'], 'device_ms', d['tpot_ms']['device_p50'])"
  done > $out/${R}_batch_sweep.txt
  for c in 128 512 2048 4096 8192 10000; do
    timeout 250 python bench.py --ctx $c --steps 32 --cpu-steps 0 --ttft-iters 2 --profile-iters 0 --ttft10k-iters 0 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('ctx', d['config']['ctx'], 'tok/s', d['value'], 'device_ms', d['tpot_ms']['device_p50'], 'ttft_ms', d['ttft_ms']['p50'])"
  done > $out/${R}_context_sweep.txt
  timeout 250 python bench.py --model qwen3-8b --steps 64 --cpu-steps 0 --profile-iters 0 --ttft10k-iters 0 2>/dev/null | tail -1 > $out/${R}_qwen3_8b_greedy.json
  timeout 250 python bench.py --model qwen3-8b --sampling topk_topp --steps 64 --cpu-steps 0 --ttft-iters 1 --profile-iters 0 --ttft10k-iters 0 2>/dev/null | tail -1 > $out/${R}_qwen3_8b_topk_topp.json
  timeout 250 python bench.py --model qwen3.5-4b --cpu-steps 0 2>/dev/null | tail -1 > $out/${R}_qwen35_4b_chained_bench.json
  cd /tmp
  kt ${R}_batch16 --batch 16 --steps 32 --cpu-steps 0 --ttft-iters 1 --profile-iters 0 --ttft10k-iters 0
  kt ${R}_batch32 --batch 32 --steps 32 --cpu-steps 0 --ttft-iters 1 --profile-iters 0 --ttft10k-iters 0
fi
ls -la $out
