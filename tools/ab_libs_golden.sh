#!/bin/bash
# A/B on the GPU box: for each variant library (file names under bendy_tracer_amd/), the golden parity tests (bit-exactness
# guard) and tools/time_workloads.py.  "libbendy_hip.so" = the freshly built default.
cd $GRAFT_REPO_ROOT
cp bendy_tracer_amd/libbendy_hip.so /tmp/base.so
for v in "$@"; do
  echo "=== $v"
  if [ "$v" = "libbendy_hip.so" ]; then cp /tmp/base.so bendy_tracer_amd/libbendy_hip.so; else cp bendy_tracer_amd/$v bendy_tracer_amd/libbendy_hip.so; fi
  timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "gpu_matches_golden or random_scenes_bit_exact" 2>&1 | tail -2
  timeout -k 10 300 python tools/time_workloads.py 2>&1 | grep -v amdgpu.ids
done
cp /tmp/base.so bendy_tracer_amd/libbendy_hip.so
