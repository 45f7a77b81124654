#!/bin/bash
# same-box A/B of kernel variants on C3 (and C5's depth): tools/gpu_ab_c3.sh <tag> lib...   (first a parity check of every variant)
cd $GRAFT_REPO_ROOT
O=gpurun_out/$1; mkdir -p $O; shift
for v in "$@"; do
  echo "== parity $v" | tee -a $O/ab_c3.log
  timeout -k 10 300 bash tools/run_with_lib.sh $v python -m pytest tests -m gpu -x -q -k "c3 or golden or random_scenes" 2>&1 | tail -1 | tee -a $O/ab_c3.log
done
for round in 1 2 3; do
for v in libbendy_hip.so "$@"; do
  echo "== $v" | tee -a $O/ab_c3.log
  BT_ONLY=${BT_ONLY:-scene} timeout -k 10 100 bash tools/run_with_lib.sh $v python tools/time_c3.py 40 2>&1 | grep -v "amdgpu.ids\|same file" | tee -a $O/ab_c3.log
done; done
