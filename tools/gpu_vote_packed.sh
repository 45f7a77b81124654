#!/bin/bash
# phase-vote wait in packed sphere launches.  usage: tools/gpu_vote_packed.sh <tag>
cd $GRAFT_REPO_ROOT
O=gpurun_out/$1; mkdir -p $O
for f in 768x512 1920x1080; do
  BT_ONLY=scene,volume BT_FRAME=$f BT_MODES=auto,v0,v1,v2,v3,v4,v6,auto BT_T=1,4,8 timeout -k 10 200 python tools/time_shallow.py 2>&1 | grep -v amdgpu.ids | tee -a $O/vote_packed.log
done
