#!/bin/bash
# C2 with smaller workgroups x slices.  usage: tools/ab_c2_wg.sh
cd $GRAFT_REPO_ROOT
export BT_ONLY=cornell2,cornell
cp bendy_tracer_amd/libbendy_hip.so /tmp/base.so
for lib in base t128 t64; do
  [ $lib = base ] || cp bendy_tracer_amd/libbendy_hip_$lib.so bendy_tracer_amd/libbendy_hip.so
  for s in 1 2 4 8; do echo "=== $lib BT_SLICES=$s"; BT_SLICES=$s python tools/time_workloads.py 2>&1 | grep -v amdgpu.ids; done
  cp /tmp/base.so bendy_tracer_amd/libbendy_hip.so
done
