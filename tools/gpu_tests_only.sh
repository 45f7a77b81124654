#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-r02}; mkdir -p $O
timeout -k 10 1100 python -m pytest tests -m gpu -x -q ${@:2} > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -15 $O/pytest_gpu.log
