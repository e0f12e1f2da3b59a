#!/bin/bash
repo=${GRAFT_REPO_ROOT:-/root/repo}
cd $repo
mkdir -p gpurun_out/profiles_new
timeout 600 python tools/op_report.py --iters 10 2>&1 | grep -v amdgpu.ids > gpurun_out/profiles_new/r2_op_report_cold.txt
tail -70 gpurun_out/profiles_new/r2_op_report_cold.txt
