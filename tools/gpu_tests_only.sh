#!/bin/bash
# pytest -m gpu only.  usage: tools/gpu_tests_only.sh <tag> [pytest -k expression]
cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-tests}; mkdir -p $O
timeout -k 10 1000 python -m pytest tests -m gpu -x -q ${2:+-k "$2"} > $O/pytest_gpu.log 2>&1; rc=$?; tail -5 $O/pytest_gpu.log; exit $rc
