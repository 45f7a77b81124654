#!/bin/bash
# Full GPU validation: pytest -m gpu, smoke, a bench line.  usage: tools/gpu_check.sh <tag>
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
TAG=${1:-r02}; O=gpurun_out/$TAG; mkdir -p $O
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -5 $O/pytest_gpu.log
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 $O/smoke.log
timeout -k 10 600 python bench.py --steps 20 --warmup 5 > $O/bench.log 2>$O/bench.err; echo "bench rc=$?"; cat $O/bench.log; tail -3 $O/bench.err
