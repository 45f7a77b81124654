#!/bin/bash
# lane statistics (developer build -DBT_LANESTAT as libbendy_hip_ls.so): deep workloads and small launches.  usage: tools/gpu_lanestat.sh <tag>
cd $GRAFT_REPO_ROOT
O=gpurun_out/$1; mkdir -p $O
timeout -k 10 200 bash tools/run_with_lib.sh libbendy_hip_ls.so python tools/time_workloads.py 2>&1 | grep -v "amdgpu.ids\|same file" | awk '/bt lanes/{last=$0; next} {if (last!="") print last; last=""; print}' | tee $O/lanestat.log
BT_FRAME=768x512 BT_MODES=auto BT_T=4 timeout -k 10 100 bash tools/run_with_lib.sh libbendy_hip_ls.so python tools/time_shallow.py 2>&1 | grep -v "amdgpu.ids\|same file" | awk '/bt lanes/{last=$0; next} {if (last!="") print last; last=""; print}' | tee -a $O/lanestat.log
