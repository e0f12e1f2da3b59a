#!/bin/bash
repo=${GRAFT_REPO_ROOT:-/root/repo}
cd $repo
timeout 600 python -m pytest tests/test_gpu_model.py tests/test_gpu_tp2.py -m gpu -x -q -k "echo or tp2" 2>&1 | tail -3
bash tools/gpu_refresh_profiles.sh r2 b 2>&1 | tail -3
