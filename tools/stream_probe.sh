#!/bin/bash
cd $GRAFT_REPO_ROOT
[ -n "$BT_LIB" ] && cp bendy_tracer_amd/$BT_LIB bendy_tracer_amd/libbendy_hip.so
for c in "$@"; do timeout -k 5 25 python tools/stream_probe.py $c 2>&1 | grep -v amdgpu.ids || { echo "FAILED/TIMEOUT: $c"; }; done
