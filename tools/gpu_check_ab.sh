#!/bin/bash
# full -m gpu suite, then the same-box A/B of tools/gpu_ab_tree.sh
repo=${GRAFT_REPO_ROOT:-/root/repo}
cd $repo
bash tools/gpu_tests.sh
bash tools/gpu_ab_tree.sh "${1:-1 16}" ${2:-96}
