#!/bin/bash
# Developer tool: section shares of wave cycles on the BASELINE workloads.  Needs the -DBT_PROFILE build:
#   make -C bendy_tracer_amd/csrc profile     (here, before gpurun; the .so travels with the snapshot)
cd $GRAFT_REPO_ROOT
cp bendy_tracer_amd/libbendy_hip.so /tmp/base.so
cp bendy_tracer_amd/libbendy_hip_profile.so bendy_tracer_amd/libbendy_hip.so
python tools/time_workloads.py 2>&1 | grep -v amdgpu.ids | awk '/bt profile/{last=$0; next} {if (last!="") print last; last=""; print}'
cp /tmp/base.so bendy_tracer_amd/libbendy_hip.so
