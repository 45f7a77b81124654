#!/bin/bash
# quick same-box A/B on the BASELINE workloads: tools/gpu_ab_quick.sh <tag> variant.so [pytest -k expression]
cd $GRAFT_REPO_ROOT
O=gpurun_out/$1; mkdir -p $O
if [ -n "$3" ]; then timeout -k 10 600 python -m pytest tests -m gpu -x -q -k "$3" > $O/pytest_gpu.log 2>&1; rc=$?; tail -3 $O/pytest_gpu.log; if [ $rc -ne 0 ]; then exit 1; fi; fi
for v in libbendy_hip.so $2 libbendy_hip.so $2; do
  echo "== $v" | tee -a $O/ab_quick.log; timeout -k 10 150 bash tools/run_with_lib.sh $v python tools/time_workloads.py 2>&1 | grep -v "amdgpu.ids\|same file" | tee -a $O/ab_quick.log
done
