#!/bin/bash
# first GPU call of round 2: instruction issue costs, class counters of the three scene classes, baseline timings
set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r02a; mkdir -p $O
timeout -k 10 300 tools/valu_microbench $O/valu_microbench.json > $O/valu_microbench.log 2>&1
echo microbench done
for wl in C3 cornell1080 C4 C2; do
  timeout -k 10 300 python3 tools/pmc_collect.py --workload $wl --out $O/pmc_$wl.json > $O/pmc_$wl.log 2>&1 || { echo "pmc $wl FAILED"; tail -5 $O/pmc_$wl.log; }
  echo pmc $wl done
done
timeout -k 10 300 python3 tools/time_workloads.py > $O/time_workloads.log 2>&1
cat $O/time_workloads.log
