#!/bin/bash
# end-of-round check: full -m gpu suite, smoke(), default bench line, sweeps (profiles part b)
repo=${GRAFT_REPO_ROOT:-/root/repo}
cd $repo
bash tools/gpu_tests.sh
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
mkdir -p gpurun_out/profiles_new
timeout 300 python bench.py > gpurun_out/profiles_new/r2_bench_default_run.json 2>/dev/null
python -c "import json;d=json.load(open('gpurun_out/profiles_new/r2_bench_default_run.json'));print(d['value'],d['tpot_ms'],d['ttft_ms'],d['roofline']['frac'],d['step_roofline']['frac_of_8TBps'])"
bash tools/gpu_refresh_profiles.sh r2 b > /dev/null 2>&1
cat gpurun_out/profiles_new/r2_context_sweep.txt gpurun_out/profiles_new/r2_batch_sweep.txt
