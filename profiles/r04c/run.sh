#!/bin/bash
# round 3, step c: the path pool (end-game compaction + march stack).  Tests first (every GPU step under its own timeout: a
# pool protocol bug would show up as a hang), then A/B of the pool's knobs on the BASELINE workloads.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r04c; mkdir -p $O
timeout -k 10 120 python -m pytest tests -m gpu -x -q -k "path_pool_is_scheduling_only or shallow_launches or golden" > $O/pytest_pool.log 2>&1; rc=$?; echo "pool tests rc=$rc"; tail -5 $O/pytest_pool.log
[ $rc -ne 0 ] && exit 1
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -5 $O/pytest_gpu.log
[ $rc -ne 0 ] && exit 1
for eg in 0 8 16 24 32 48; do
  echo "== BT_END_GAME=$eg BT_MARCH_POOL=0"; BT_END_GAME=$eg BT_MARCH_POOL=0 timeout -k 10 120 python tools/time_workloads.py 2>&1 | grep -v amdgpu.ids | tee -a $O/ab_end_game.log
done
for mp in "64 16" "64 32" "128 16" "128 32" "128 64" "256 32" "256 96"; do set -- $mp
  echo "== BT_END_GAME=16 BT_MARCH_POOL=$1 BT_MARCH_ENTER=$2"; BT_ONLY=volume,cloud BT_END_GAME=16 BT_MARCH_POOL=$1 BT_MARCH_ENTER=$2 timeout -k 10 120 python tools/time_workloads.py 2>&1 | grep -v amdgpu.ids | tee -a $O/ab_march_pool.log
done
echo "== default"; timeout -k 10 120 python tools/time_workloads.py 2>&1 | grep -v amdgpu.ids | tee $O/time_default.log
