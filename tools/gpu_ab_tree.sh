#!/bin/bash
# Same-box A/B of two source trees (box-to-box spread on the pool is 2-3 %, larger than most kernel changes):
#   git worktree add -f _ab <commit> && (cd _ab && python -m pegainfer_amd.build)      # once, in the container
#   gpurun -- 'bash tools/gpu_ab_tree.sh "1 16" 96'                                     # batches, steps
# prints device ms per step for _ab (base) and the working tree (new), alternating twice.
repo=${GRAFT_REPO_ROOT:-/root/repo}
batches=${1:-"1 16"}; steps=${2:-96}
cd $repo
sw() { # dir label
  for b in $batches; do
    (cd $1 && timeout 200 python bench.py --batch $b --steps $steps --cpu-steps 0 --ttft-iters 1 --profile-iters 0 --ttft10k-iters 0 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$2 bs', d['config']['batch_per_gpu'], 'tok/s', d['value'], 'device_ms', d['tpot_ms']['device_p50'], 'ttft_ms', d['ttft_ms']['p50'])")
  done
}
sw _ab base; sw . new; sw _ab base; sw . new
