#!/bin/bash
# Block queue with smaller workgroups: variant libraries (BT_WG_THREADS=128 / 64) x BT_SLICES.  usage: tools/ab_wg_threads.sh
cd $GRAFT_REPO_ROOT
cp bendy_tracer_amd/libbendy_hip.so /tmp/base.so
run() { echo "=== $1 BT_SLICES=$2"; if [ "$2" = "-" ]; then python tools/time_workloads.py 2>&1 | grep -v amdgpu.ids; else BT_SLICES=$2 python tools/time_workloads.py 2>&1 | grep -v amdgpu.ids; fi; }
run base -
cp bendy_tracer_amd/libbendy_hip_t128.so bendy_tracer_amd/libbendy_hip.so
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "gpu_matches_golden" 2>&1 | tail -1
for s in - 8 16; do run t128 $s; done
cp bendy_tracer_amd/libbendy_hip_t64.so bendy_tracer_amd/libbendy_hip.so
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "gpu_matches_golden" 2>&1 | tail -1
for s in 8 16 32; do run t64 $s; done
cp /tmp/base.so bendy_tracer_amd/libbendy_hip.so
run base -
