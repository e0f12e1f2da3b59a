mkdir -p gpurun_out
out=gpurun_out/r4_weights_nt_ab.txt; : > $out
for nt in 0 1; do
  export PEGAINFER_WEIGHTS_NT=$nt
  for T in 32 64 128 256; do
    python tools/bench_prefill_gemm.py $T 12 2>&1 | grep -v amdgpu | sed "s/^/[NT=$nt] /" >> $out
  done
  python tools/ttft_probe.py 32 64 128 256 2>&1 | grep TTFT | sed "s/^/[NT=$nt] /" >> $out
  for b in 32 64; do
    timeout 200 python bench.py --batch $b --steps 48 --cpu-steps 0 --ttft-iters 1 --profile-iters 0 --ttft10k-iters 0 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('[NT=$nt] bs', d['config']['batch_per_gpu'], 'tok/s', d['value'], 'device_ms', d['tpot_ms']['device_p50'])" >> $out
  done
done
cat $out
