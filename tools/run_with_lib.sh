#!/bin/bash
# Developer tool: run a command on the GPU box with a variant library in place of libbendy_hip.so.
# usage: tools/run_with_lib.sh libbendy_hip_<name>.so <command ...>
cd $GRAFT_REPO_ROOT
lib=$1; shift
cp bendy_tracer_amd/libbendy_hip.so /tmp/base.so
cp bendy_tracer_amd/$lib bendy_tracer_amd/libbendy_hip.so
"$@"
rc=$?
cp /tmp/base.so bendy_tracer_amd/libbendy_hip.so
exit $rc
