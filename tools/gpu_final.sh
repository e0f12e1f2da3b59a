#!/bin/bash
# Run on the GPU box (via gpurun): the end-of-round evidence set into gpurun_out/final/ - full GPU suite, the default
# bench line, batch / context / short-prompt sweeps, kernel stats of the three headline workloads.  R = round tag.
R=${1:-r4}
repo=${GRAFT_REPO_ROOT:-/root/repo}
out=$repo/gpurun_out/final
mkdir -p $out
cd $repo
timeout 1200 python -m pytest tests -m gpu -q > $out/${R}_gpu_suite.log 2>&1
tail -3 $out/${R}_gpu_suite.log
timeout 400 python bench.py > $out/${R}_bench_default_run.json 2> $out/${R}_bench_default_run.err
for b in 2 4 8 16 32 64; do
  timeout 200 python bench.py --batch $b --steps 48 --cpu-steps 0 --ttft-iters 1 --profile-iters 0 --ttft10k-iters 0 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('bs', d['config']['batch_per_gpu'], 'tok/s', d['value'], 'device_ms', d['tpot_ms']['device_p50'])"
done > $out/${R}_batch_sweep.txt
for c in 128 512 2048 4096 8192 10000; do
  timeout 250 python bench.py --ctx $c --steps 32 --cpu-steps 0 --ttft-iters 2 --profile-iters 0 --ttft10k-iters 0 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('ctx', d['config']['ctx'], 'tok/s', d['value'], 'device_ms', d['tpot_ms']['device_p50'], 'ttft_ms', d['ttft_ms']['p50'])"
done > $out/${R}_context_sweep.txt
python tools/ttft_probe.py 1 4 8 16 17 32 64 100 128 256 512 1024 2>&1 | grep TTFT > $out/${R}_ttft_sweep.txt
cd /tmp && export TMPDIR=/tmp
kt() {  # name, bench args...
  local name=$1; shift
  rm -rf /tmp/prof_$name
  timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_$name -o run -- python $repo/bench.py "$@" > /tmp/prof_$name.log 2>&1
  python $repo/tools/rocpd_stats.py "$(find /tmp/prof_$name -name '*.db' | head -1)" $out/${name}_kernel_stats.csv > /dev/null
  grep '^{' /tmp/prof_$name.log | tail -1 > $out/${name}_bench_under_rocprof.json
}
kt ${R}_fused_decode_mode1 --steps 64 --cpu-steps 0 --ttft-iters 5 --profile-iters 0 --ttft10k-iters 0
kt ${R}_batch32 --batch 32 --steps 32 --cpu-steps 0 --ttft-iters 1 --profile-iters 0 --ttft10k-iters 0
kt ${R}_ctx10000 --ctx 10000 --steps 8 --cpu-steps 0 --ttft-iters 2 --profile-iters 0 --ttft10k-iters 0
ls -la $out
