#!/bin/bash
# Fuzz on the GPU box: random launch shapes and random scenes against the oracle.  usage: tools/gpu_fuzz.sh <tag> [shapes] [scenes]
cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-fuzz}; mkdir -p $O
timeout -k 10 500 python tools/fuzz_shapes.py ${2:-400} 11 > $O/fuzz_launch_shapes.log 2>&1; echo "shapes rc=$?"; tail -3 $O/fuzz_launch_shapes.log
timeout -k 10 600 python tools/fuzz_sweep.py 5000 ${3:-8000} > $O/fuzz_sweep.log 2>&1; echo "sweep rc=$?"; tail -3 $O/fuzz_sweep.log
