#!/bin/bash
# The N > 1 code path on a one-GPU box: RCCL ABI at world 1, gloo rehearsals with 2 and 3 ranks on cuda:0 (weak and strong
# scaling, verified against a one-rank render).  usage: tools/gpu_multi_rehearsal.sh <tag>
cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-r02}; mkdir -p $O
export MASTER_ADDR=127.0.0.1
timeout -k 10 300 python bench.py --force-dist --backend rccl-abi --steps 5 --warmup 2 --no-cpu-baseline --verify > $O/bench_rccl_abi_world1.log 2>$O/bench_rccl_abi_world1.err; echo "rccl-abi world 1 rc=$?"; cat $O/bench_rccl_abi_world1.log | cut -c1-400
timeout -k 10 300 python bench.py --force-dist --backend nccl --steps 5 --warmup 2 --no-cpu-baseline --verify > $O/bench_nccl_world1.log 2>&1; echo "nccl world 1 rc=$?"
for cfg in "2 weak C3" "3 strong C3" "2 strong C2"; do set -- $cfg
  timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node $1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus $1 --backend gloo --single-device --scaling $2 --workload $3 --steps 4 --warmup 1 --no-cpu-baseline --verify > $O/bench_gloo$1_$2_$3.log 2>&1; echo "gloo $cfg rc=$?"; grep -o '"value": [0-9.]*\|"verified_vs_single_rank": [a-z]*\|"final_gather_only": {[^}]*}\|"scaling": "[a-z]*"' $O/bench_gloo$1_$2_$3.log | tr '\n' ' '; echo
done
