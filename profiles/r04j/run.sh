#!/bin/bash
# block size (bt_tuning.slices) on small frames, re-measured with round 3's kernel: is bt_api.cpp's pick() still right?
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04j; mkdir -p $O
for f in 512x512 768x512 1280x720; do
  echo "== $f"; BT_MODES=auto,s1,s2,s4,s8,s16 BT_T=1,2,4,8,16,32,64 BT_FRAME=$f timeout -k 10 300 python tools/time_shallow.py 2>&1 | grep -v amdgpu.ids | tee $O/time_blocksize_$f.log
done
