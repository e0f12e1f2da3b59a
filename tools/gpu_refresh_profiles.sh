#!/bin/bash
# Run on the GPU box (via gpurun): regenerate every file under profiles/ for this round into gpurun_out/profiles_new/.
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
pmc() {  # counter
  rm -rf /tmp/pmc_$1
  timeout 300 rocprofv3 --pmc $1 --kernel-trace -d /tmp/pmc_$1 -o run -- python $repo/bench.py --steps 8 --cpu-steps 0 --ttft-iters 1 > /tmp/pmc_$1.log 2>&1
  python $repo/tools/rocpd_pmc.py "$(find /tmp/pmc_$1 -name '*.db' | head -1)" $out/r1_fused_pmc_$1.csv
}
kt r1_fused_decode_mode1 --steps 64 --cpu-steps 0
kt r1_baseline_decode_mode0 --steps 64 --cpu-steps 0 --decode-mode 0
kt r1_qwen35_4b --model qwen3.5-4b --steps 64
pmc FETCH_SIZE
pmc WRITE_SIZE
cd $repo
timeout 300 python bench.py > $out/r1_bench_default_run.json 2>/dev/null
for b in 2 4 8 16 32 64; do
  timeout 200 python bench.py --batch $b --steps 48 --cpu-steps 0 --ttft-iters 1 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('bs', d['config']['batch_per_gpu'], 'tok/s', d['value'], 'device_ms', d['tpot_ms']['device_p50'])"
done > $out/r1_batch_sweep.txt
for c in 128 512 2048 4096 8192 10000; do
  timeout 250 python bench.py --ctx $c --steps 32 --cpu-steps 0 --ttft-iters 2 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('ctx', d['config']['ctx'], 'tok/s', d['value'], 'device_ms', d['tpot_ms']['device_p50'], 'ttft_ms', d['ttft_ms']['p50'])"
done > $out/r1_context_sweep.txt
timeout 250 python bench.py --model qwen3-8b --steps 64 --cpu-steps 0 2>/dev/null | tail -1 > $out/r1_qwen3_8b_greedy.json
timeout 250 python bench.py --model qwen3-8b --sampling topk_topp --steps 64 --cpu-steps 0 --ttft-iters 1 2>/dev/null | tail -1 > $out/r1_qwen3_8b_topk_topp.json
timeout 250 python bench.py --model qwen3.5-4b 2>/dev/null | tail -1 > $out/r1_qwen35_4b_bench.json
ls -la $out
