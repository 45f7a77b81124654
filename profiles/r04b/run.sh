#!/bin/bash
# round 3, step b: pruned kernel + hoisted block geometry + scalar phase vote + flow queue.  Tests, default timings, Philox-7 A/B,
# both queues on shallow launches.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r04b; mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -8 $O/pytest_gpu.log
[ $rc -ne 0 ] && exit 1
echo "== default"; timeout -k 10 200 python tools/time_workloads.py 2>&1 | grep -v amdgpu.ids | tee $O/time_default.log
echo "== philox7 (timing only: its pixels differ from the contract's)"; timeout -k 10 200 bash tools/run_with_lib.sh libbendy_hip_philox7.so python tools/time_workloads.py 2>&1 | grep -v amdgpu.ids | tee $O/ab_philox7.log
for f in 1920x1080 768x512 512x512; do
  echo "== shallow $f"; BT_FRAME=$f timeout -k 10 400 python tools/time_shallow.py 2>&1 | grep -v amdgpu.ids | tee $O/time_shallow_$f.log
done
