#!/bin/bash
repo=${GRAFT_REPO_ROOT:-/root/repo}
cd $repo
mkdir -p gpurun_out
ab() {  # label, args, env...
  label=$1; shift; args=$1; shift
  env "$@" timeout 200 python bench.py $args --steps 64 --cpu-steps 0 --ttft-iters 1 --profile-iters 0 --ttft10k-iters 0 2>/dev/null | tail -1 | \
    python -c "import sys,json;d=json.loads(sys.stdin.read());print('$label','tok/s',d['value'],'device_ms',d['tpot_ms']['device_p50'],{k:v['us'] for k,v in d['gemv_sites'].items()},'fused gate_up',d['roofline']['avg_launch_us'])"
}
for i in 1 2; do
  ab "contig=0 " "" PEGAINFER_GEMV_CONTIG=0
  ab "contig=1 " "" PEGAINFER_GEMV_CONTIG=1
done 2>&1 | tee gpurun_out/r3s_contig_ab.txt
ab "contig=0 bs2" "--batch 2" PEGAINFER_GEMV_CONTIG=0 | tee -a gpurun_out/r3s_contig_ab.txt
ab "contig=1 bs2" "--batch 2" PEGAINFER_GEMV_CONTIG=1 | tee -a gpurun_out/r3s_contig_ab.txt
timeout 900 python -m pytest tests/test_gpu_fused.py tests/test_gpu_real_dims.py tests/test_gpu_ops.py -m gpu -q -x --tb=short > gpurun_out/pytest_r3s.log 2>&1
tail -3 gpurun_out/pytest_r3s.log
