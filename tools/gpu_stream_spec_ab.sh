# Run on the GPU box (via gpurun): A/B of one stream-GEMM switch (default PEGAINFER_STREAM_SPLITK), stand-alone layer GEMMs and
# in the pipeline (TTFT at short prompts, batched decode).  usage: bash tools/gpu_stream_spec_ab.sh [ENV_NAME]
var=${1:-PEGAINFER_STREAM_SPLITK}
mkdir -p gpurun_out
out=gpurun_out/r4_stream_ab_$var.txt; : > $out
for v in 0 1 0 1; do
  export $var=$v
  for T in 32 64; do
    python tools/bench_prefill_gemm.py $T 12 2>&1 | grep -v "amdgpu\|ragged" | sed "s/^/[$var=$v] /" >> $out
  done
  python tools/ttft_probe.py 17 32 64 2>&1 | grep TTFT | sed "s/^/[$var=$v] /" >> $out
  for b in 32 64; do
    timeout 200 python bench.py --batch $b --steps 48 --cpu-steps 0 --ttft-iters 1 --profile-iters 0 --ttft10k-iters 0 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('[$var=$v] bs', d['config']['batch_per_gpu'], 'tok/s', d['value'], 'device_ms', d['tpot_ms']['device_p50'])" >> $out
  done
done
cat $out
