#!/bin/bash
# usage: tools/gpu_tests_only.sh <tag> [pytest args...]
cd $GRAFT_REPO_ROOT
TAG=${1:-r02}; shift
O=gpurun_out/$TAG; mkdir -p $O
timeout -k 10 1100 python -m pytest tests -m gpu -x -q "$@" > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -15 $O/pytest_gpu.log
