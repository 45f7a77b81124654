#!/bin/bash
# Profiling recipe for the GPU box (run via gpurun):  tools/profile.sh <tag> [commit]
#   1. rocprofv3 --kernel-trace --stats over `python3 bench.py` (the bench's own command line)  -> kernel_stats.csv
#   2. tools/pmc_collect.py for every workload: --pmc passes, one counter group per run (gpurun refuses --pmc together
#      with API traces)                                                                       -> pmc_<workload>.json
#   3. profiles/pmc_live.json = the summaries bench.py quotes when it cannot measure live, stamped with the source hash
#      (tools/pmc_collect.py source_hash) and, when given, the commit (the GPU box has no .git: pass `git rev-parse HEAD`)
# Everything lands in gpurun_out/prof_<tag>/; copy what should be judged into profiles/<tag>/.
export TMPDIR=/tmp
TAG=${1:-r02}; COMMIT=${2:-unknown}
cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-pmc --no-other-configs > $OUT/bench_under_trace.log 2>$OUT/trace.err || tail -3 $OUT/trace.err
find $OUT/trace -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats.csv \;
for wl in C3 C2 C4 cornell1080 cloud1080; do
  python3 tools/pmc_collect.py --workload $wl --passes fetch,write,sq,classes,waits --commit $COMMIT --out $OUT/pmc_$wl.json > /dev/null 2>$OUT/pmc_$wl.err || { echo "pmc $wl failed"; tail -3 $OUT/pmc_$wl.err; }
done
# one rank's launch of the N-rank weak-scaling jobs (bench.py --gpus N: 1/N of the tiles at 64 N spp) -> keys C3_shard2 / 4 / 8
for n in 2 4 8; do
  python3 tools/pmc_collect.py --workload C3 --shard 0,$n --spp $((64 * n)) --calls 3 --passes fetch,write,sq,classes,waits --commit $COMMIT --out $OUT/pmc_C3_shard$n.json > /dev/null 2>$OUT/pmc_C3_shard$n.err || { echo "pmc C3 shard $n failed"; tail -3 $OUT/pmc_C3_shard$n.err; }
done
python3 - <<PY
import json, glob, os
out = {}
for f in sorted(glob.glob("$OUT/pmc_*.json")):
    d = json.load(open(f)); out[d["workload"] + (f"_shard{d['shard']['world']}" if d.get("shard") else "")] = d
json.dump(out, open("$OUT/pmc_live.json", "w"), indent=1)
for w, d in out.items():
    x = d["derived"]
    print(f"{w:14s} sha {d['source_sha']} kernel {d['cli']['kernel_ms_under_profiler']:.3f} ms (under the profiler)  VALU/SIMD-cycle {x['valu_per_simd_cycle']:.3f}  lanes {x['lanes_active']:.3f}  "
          f"mixed-cost {x.get('valu_issue_mixed_frac', 0):.3f} (pure-stream bound {x.get('valu_issue_weighted_frac', 0):.3f})  scalar/CU-cycle {x['scalar_per_cu_cycle']:.3f}  HBM {x['hbm_bytes']/1e9:.3f} GB")
PY
head -5 $OUT/kernel_stats.csv
