#!/bin/bash
# round 3 end-of-round check: full -m gpu suite, smoke(), the default bench line
repo=${GRAFT_REPO_ROOT:-/root/repo}
cd $repo
mkdir -p gpurun_out
t0=$(date +%s)
timeout 1500 python -m pytest tests -m gpu -q --tb=short > gpurun_out/pytest_gpu.log 2>&1
echo "pytest rc $?" >> gpurun_out/pytest_gpu.log
grep -E "passed|failed|error|rc |^FAILED|^ERROR" gpurun_out/pytest_gpu.log | tail -12
echo "suite took $(( $(date +%s) - t0 )) s"
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 600 python bench.py > gpurun_out/r3_bench_default_run.json 2> gpurun_out/r3_bench_default_run.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/r3_bench_default_run.json'))
print('value',d['value'],'tpot',d['tpot_ms'],'ttft',d['ttft_ms']['p50'],'10k',d['ttft_ms_10000']['p50'],'roofline',d['roofline']['frac'],d['roofline']['avg_launch_us'],d['roofline']['traffic'],'step',d['step_roofline']['frac_of_8TBps'])
PY
