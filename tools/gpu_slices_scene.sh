#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/$1; mkdir -p $O
for s in 4 8 16 32 4 8; do
  echo "== slices $s" | tee -a $O/slices_scene.log
  BT_SLICES=$s BT_ONLY=scene timeout -k 10 100 python tools/time_c3.py 40 2>&1 | grep -v amdgpu.ids | tee -a $O/slices_scene.log
done
