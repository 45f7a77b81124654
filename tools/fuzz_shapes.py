#!/usr/bin/env python3
"""Developer tool (GPU box): launch shapes against the oracle -- random frame sizes (up to 1920x1080), 1 ... 48 samples,
Subsample 1 ... 3, every Output mode, full frames and rank shards, on the three scene classes; whatever bt_api.cpp picks (slices, launches), and
every pinned block shape, must give the oracle's bits.  usage: python3 tools/fuzz_shapes.py [n_cases] [seed]"""
import json
import os
import random
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
import torch

import bendy_tracer_amd as bendy
import bt_oracle_py as oracle
from helpers import gpu_scene, oracle_render, unshard_numpy

n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 7)
bad = 0
seen = {}
t0 = time.time()
for case in range(n):
    name = rng.choice(["scene", "cornell2", "volume", "cornell", "cloud"])
    big = case % 25 == 0
    w, h = (1920, 1080) if big else (rng.randint(17, 700), rng.randint(9, 500))
    spp = rng.choice([1, 1, 2, 3, 5, 8, 13, 16, 24, 48]) if not big else rng.choice([1, 2, 4])
    sub = rng.choice([0, 0, 2, 3]) if spp <= 8 else 0
    world = rng.choice([1, 1, 1, 2, 3])
    output = rng.choice([0, 0, 0, 1, 2, 3])                       # Output::{Full, Albedo, Normal, Depth}
    pin = rng.choice([None, None, None, 1, 2, 4, 8, 16, 32])      # bt_tuning.slices: every block shape, not only the automatic one
    packed = rng.choice([-1, 0, 1, 2, 2])                            # bt_tuning.packed: several blocks behind one queue
    tuning = {"packed": packed}
    if pin:
        tuning["slices"] = pin
    sc, cam = gpu_scene(bendy, name, w, h, tuning=tuning)
    tr = bendy.Tracer.with_config(bendy.Config(chunks_x=8, chunks_y=4, output=bendy.Output(output)))
    rc = bendy.RenderConfig.with_samples_subsample(spp, bendy.Subsample(sub)) if sub else bendy.RenderConfig.with_samples(spp)
    if world == 1:
        buf = bendy.Buffer.new(w, h)
        tr.render(sc, cam, rc, buf, seed=0x5EED)
        torch.cuda.synchronize()
        got = buf.numpy()
    else:
        shards = []
        for r in range(world):
            s = bendy.new_shard(w, h, world)
            tr.render_shard(sc, cam, rc, s, w, h, r, world, seed=0x5EED)
            shards.append(s)
        out = bendy.Buffer.new(w, h)
        bendy.unshard(torch.cat(shards), out, world)
        torch.cuda.synchronize()
        got = out.numpy()
    st = sc.last_stats()
    it, seg = oracle_render(oracle, name, w, h, spp, n=sub, output=output, recursive=0, threads=16)
    ok = np.array_equal(got[..., :3], it[..., :3], equal_nan=True)
    key = (st.slices, st.launches, packed, st.packed)
    seen[key] = seen.get(key, 0) + 1
    if not ok:
        bad += 1
        print(f"MISMATCH case {case}: {name} {w}x{h} spp {spp} sub {sub} world {world} output {output} slices {st.slices}", flush=True)
    if case % 50 == 49:
        print(f"... {case + 1} cases, {bad} mismatches, {time.time() - t0:.0f} s", flush=True)
print("slices / launches / packed knob / packed launch seen:", dict(sorted(seen.items())))
print(f"launch shapes: {n - bad} of {n} cases bit-identical to the oracle")
sys.exit(1 if bad else 0)
