#!/bin/bash
# Drain compaction of the packed rect build: cadence / pool variants on small Cornell-box launches.  usage: tools/gpu_ab_drain.sh <tag> lib...
cd $GRAFT_REPO_ROOT
O=gpurun_out/$1; mkdir -p $O; shift
for v in libbendy_hip.so "$@" libbendy_hip.so; do
  echo "== $v" | tee -a $O/ab_drain.log
  for f in 512x512 768x512; do
    BT_ONLY=cornell2 BT_FRAME=$f BT_MODES=p,c BT_T=1,2,4,8,16,32 timeout -k 10 120 bash tools/run_with_lib.sh $v python tools/time_shallow.py 2>&1 | grep -v "amdgpu.ids\|same file" | tee -a $O/ab_drain.log
  done
done
