#!/bin/bash
# round 3, step e: end-game compaction in a second copy of the loop body.  Tests, then timings: default build (7 waves/SIMD) with
# end_game 0 / 16 / 32 / 48, the same source without any pool code (-DBT_POOL=0), and 6 waves/SIMD (80 VGPRs).
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r04e; mkdir -p $O
timeout -k 10 150 python -m pytest tests -m gpu -x -q -k "path_pool_is_scheduling_only or shallow_launches or golden" > $O/pytest_pool.log 2>&1; rc=$?; echo "pool tests rc=$rc"; tail -3 $O/pytest_pool.log
[ $rc -ne 0 ] && exit 1
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -3 $O/pytest_gpu.log
[ $rc -ne 0 ] && exit 1
echo "== nopool build"; timeout -k 10 120 bash tools/run_with_lib.sh libbendy_hip_nopool.so python tools/time_workloads.py 2>&1 | grep -v amdgpu.ids | tee $O/time_nopool.log
for eg in 0 16 32 48; do
  echo "== default build BT_END_GAME=$eg"; BT_END_GAME=$eg timeout -k 10 120 python tools/time_workloads.py 2>&1 | grep -v amdgpu.ids | tee -a $O/ab_end_game.log
done
for eg in 0 32; do
  echo "== w6 build BT_END_GAME=$eg"; BT_END_GAME=$eg timeout -k 10 120 bash tools/run_with_lib.sh libbendy_hip_w6.so python tools/time_workloads.py 2>&1 | grep -v amdgpu.ids | tee -a $O/ab_w6.log
done
