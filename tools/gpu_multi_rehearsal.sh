#!/bin/bash
# The N > 1 code path on a one-GPU box: RCCL ABI at world 1, then gloo rehearsals with 2 and 3 ranks on cuda:0 (weak and
# strong scaling) -- started exactly as a user would type them, `python bench.py --gpus N ...` from a bare shell: bench.py
# launches its own ranks as a child torchrun (tests/test_bench_launch.py), rank 0 profiles its own shard launch, every rank
# verifies one extra step against a one-rank render.  usage: tools/gpu_multi_rehearsal.sh <tag>
cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-r03}; mkdir -p $O
unset WORLD_SIZE RANK LOCAL_RANK
export MASTER_ADDR=127.0.0.1
timeout -k 10 300 python bench.py --force-dist --backend rccl-abi --steps 5 --warmup 2 --no-cpu-baseline --verify > $O/bench_rccl_abi_world1.log 2>$O/bench_rccl_abi_world1.err; echo "rccl-abi world 1 rc=$?"; cut -c1-400 $O/bench_rccl_abi_world1.log
timeout -k 10 300 python bench.py --force-dist --backend nccl --steps 5 --warmup 2 --no-cpu-baseline --verify > $O/bench_nccl_world1.log 2>&1; echo "nccl world 1 rc=$?"
for cfg in "2 weak C3" "3 strong C3" "2 strong C2"; do set -- $cfg
  timeout -k 10 500 python bench.py --gpus $1 --backend gloo --single-device --scaling $2 --workload $3 --steps 4 --warmup 1 --no-cpu-baseline > $O/bench_gloo$1_$2_$3.log 2>$O/bench_gloo$1_$2_$3.err; echo "gloo $cfg rc=$?"
  python3 - $O/bench_gloo$1_$2_$3.log <<'PY'
import json, sys
line = [l for l in open(sys.argv[1]) if l.startswith("{")][-1]
d = json.loads(line)
r = d["roofline"]
print({k: d.get(k) for k in ("value", "n_gpus", "ranks_joined", "distinct_devices", "verified_vs_single_rank", "scaling")},
      "devices", [(c["rank"], c["device"], c["pid"]) for c in d["devices"]],
      "roofline", {k: r.get(k) for k in ("frac", "issue_cost_guide_frac", "lane_weighted_frac", "hbm_measured_frac", "kernel_ms")}, r.get("pmc", {}).get("shard"))
PY
done
