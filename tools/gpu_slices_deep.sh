#!/bin/bash
# block size (bt_tuning.slices) on the deep BASELINE workloads, 40 warm launches each.  usage: tools/gpu_slices_deep.sh <tag>
cd $GRAFT_REPO_ROOT
O=gpurun_out/$1; mkdir -p $O
for s in 4 2 1 8 4; do
  echo "== slices $s" | tee -a $O/slices_deep.log
  BT_SLICES=$s BT_ONLY=${BT_ONLY:-volume,cloud,scene,cornell} timeout -k 10 200 python tools/time_c3.py 30 2>&1 | grep -v amdgpu.ids | tee -a $O/slices_deep.log
done
