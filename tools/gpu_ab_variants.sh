#!/bin/bash
# A/B of variant libraries on the BASELINE workloads + small launches.  usage: tools/gpu_ab_variants.sh <tag> lib1.so lib2.so ...
cd $GRAFT_REPO_ROOT
O=gpurun_out/$1; mkdir -p $O; shift
for v in libbendy_hip.so "$@" libbendy_hip.so; do
  echo "== $v"; timeout -k 10 150 bash tools/run_with_lib.sh $v python tools/time_workloads.py 2>&1 | grep -v "amdgpu.ids\|same file" | tee -a $O/ab_variants.log
  BT_MODES=auto BT_T=4,16 BT_FRAME=768x512 timeout -k 10 100 bash tools/run_with_lib.sh $v python tools/time_shallow.py 2>&1 | grep -v "amdgpu.ids\|same file" | tee -a $O/ab_variants.log
done
