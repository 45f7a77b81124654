#!/bin/bash
# A/B: phase-vote waits on the sphere scenes.  usage: ab_vote.sh lib.so waits...
cd $GRAFT_REPO_ROOT
cp bendy_tracer_amd/libbendy_hip.so /tmp/base.so
lib=$1; shift
[ "$lib" = "libbendy_hip.so" ] || cp bendy_tracer_amd/$lib bendy_tracer_amd/libbendy_hip.so
for w in "$@"; do echo "=== $lib BT_PHASE_VOTE=$w"; BT_ONLY=${BT_ONLY:-scene,volume,cloud} BT_PHASE_VOTE=$w python tools/time_workloads.py 2>&1 | grep -v amdgpu.ids; done
cp /tmp/base.so bendy_tracer_amd/libbendy_hip.so
