"""N > 1 path on CPU: two processes over gloo shard the frame by interleaved 16x16 tiles,
all-gather their shards and un-permute to row-major -- the host logic bench.py runs over RCCL.
Pixels come from the CPU oracle here (no GPU in this container); the GPU version of the same
flow is tests/test_gpu_parity.py::test_sharded_render_matches_full_frame."""
import os
import socket
import sys

import numpy as np
import pytest

from conftest import ROOT

WORKER = r"""
import os, sys
import numpy as np
import torch
import torch.distributed as dist
sys.path.insert(0, {root!r}); sys.path.insert(0, os.path.join({root!r}, "oracle")); sys.path.insert(0, os.path.join({root!r}, "tests"))
import bt_oracle_py as o
import bendy_tracer_amd as b
from helpers import oracle_render, shard_from_frame, unshard_numpy

rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
w, h, spp = [int(v) for v in os.environ["BT_TEST_FRAME"].split("x")]     # e.g. 72x40x2: 5 x 3 tiles, one padded tile at world 2
full, _ = oracle_render(o, "scene", w, h, spp, threads=2)
owner = b.tile_owner_map(w, h, world)
# this rank only keeps the pixels of the tiles it owns
mine = np.zeros_like(full); mine[..., 3] = 1.0
for ty in range(owner.shape[0]):
    for tx in range(owner.shape[1]):
        if owner[ty, tx] == rank:
            mine[ty*16:(ty+1)*16, tx*16:(tx+1)*16] = full[ty*16:(ty+1)*16, tx*16:(tx+1)*16]
shard = torch.from_numpy(shard_from_frame(mine, rank, world))
assert shard.numel() == b.shard_floats(w, h, world)
gathered = torch.empty(world * shard.numel(), dtype=torch.float32)
dist.all_gather_into_tensor(gathered, shard)
frame = unshard_numpy(gathered.numpy(), w, h, world)
assert np.array_equal(frame, full), "all-gather + unshard must reproduce the single-process frame"
# timing reduction used by bench.py: max over ranks
t = torch.tensor([1.0 + rank], dtype=torch.float64)
dist.all_reduce(t, op=dist.ReduceOp.MAX)
assert t.item() == float(world)
dist.barrier()
dist.destroy_process_group()
print("RANK_OK", rank)
"""


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.timeout(300)
@pytest.mark.parametrize("world,frame", [(2, "72x40x2"), (3, "100x50x2")])
def test_gloo_tile_shard_allgather(tmp_path, world, frame):
    """world 2: 15 tiles, one padded slot; world 3: 7 x 4 = 28 tiles (ragged right and bottom edge), 10 slots per rank with
    two padded ones -- the strong-scaling layout of bench.py --scaling strong on a frame that does not divide evenly."""
    import subprocess
    script = tmp_path / "worker.py"
    script.write_text(WORKER.format(root=ROOT))
    port = _free_port()
    procs = []
    for rank in range(world):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   OMP_NUM_THREADS="1", BT_TEST_FRAME=frame)
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.STDOUT, text=True))
    outs = [p.communicate(timeout=280)[0] for p in procs]
    for rank, (p, out) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, out
        assert f"RANK_OK {rank}" in out


def test_shard_roundtrip_numpy():
    from helpers import shard_from_frame, unshard_numpy
    rng = np.random.default_rng(0)
    for (w, h, world) in [(72, 40, 2), (1920 // 8, 1080 // 8, 8), (17, 33, 3), (16, 16, 4)]:
        frame = rng.random((h, w, 4), dtype=np.float32)
        shards = [shard_from_frame(frame, r, world) for r in range(world)]
        assert len({s.size for s in shards}) == 1
        back = unshard_numpy(np.concatenate(shards), w, h, world)
        assert np.array_equal(back, frame)
