# Run on the GPU box (via gpurun): stream GEMM with / without feeder waves (PEGAINFER_STREAM_SPEC) and with the qkv-sized
# matrices routed to it (PEGAINFER_STREAM_MIN_RT), in the pipeline: TTFT at short prompts and batched decode
mkdir -p gpurun_out
out=gpurun_out/r4_stream_spec_pipeline.txt; : > $out
for cfg in "0 3" "1 3" "1 2"; do
  set -- $cfg
  export PEGAINFER_STREAM_SPEC=$1 PEGAINFER_STREAM_MIN_RT=$2
  python tools/ttft_probe.py 17 32 64 128 2>&1 | grep TTFT | sed "s/^/[SPEC=$1 MIN_RT=$2] /" >> $out
  for b in 32 64; do
    timeout 200 python bench.py --batch $b --steps 48 --cpu-steps 0 --ttft-iters 1 --profile-iters 0 --ttft10k-iters 0 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('[SPEC=$1 MIN_RT=$2] bs', d['config']['batch_per_gpu'], 'tok/s', d['value'], 'device_ms', d['tpot_ms']['device_p50'])" >> $out
  done
done
cat $out
