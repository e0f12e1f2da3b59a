#!/bin/bash
# full suite, then bench.py with an env knob 1 / 0 alternating: tools/gpu_env_ab.sh KNOB "1 2" [steps]
repo=${GRAFT_REPO_ROOT:-/root/repo}
knob=$1; batches=${2:-"1"}; steps=${3:-96}
cd $repo
bash tools/gpu_tests.sh
for x in 1 0 1 0; do
  for b in $batches; do
    env $knob=$x timeout 300 python bench.py --batch $b --steps $steps --cpu-steps 0 --ttft-iters 1 --profile-iters 0 --ttft10k-iters 0 2>/dev/null | tail -1 | python -c "import sys,json;d=json.loads(sys.stdin.read());print('$knob=$x bs',$b,'tok/s',d['value'],'device_ms',d['tpot_ms']['device_p50'],d['roofline']['avg_launch_us'])"
  done
done
