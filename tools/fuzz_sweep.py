#!/usr/bin/env python3
"""Developer tool (GPU box): the fuzz of tests/test_gpu_parity.py::test_random_scenes_bit_exact over many more seeds,
larger frames and every Output mode -- HIP path vs the CPU oracle, bit for bit, segment counts included.
usage: python3 tools/fuzz_sweep.py [first_seed] [n_seeds]        (oracle = test infrastructure; this tool is a test)"""
import json
import os
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
from scene_gen import random_scene

first = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
n = int(sys.argv[2]) if len(sys.argv) > 2 else 200
bad = 0
t0 = time.time()
for seed in range(first, first + n):
    txt = random_scene(seed, n_objects=3 + seed % 12, volume_prob=0.35 if seed % 3 == 0 else 0.15)
    w, h, spp = 96 + 16 * (seed % 3), 64 + 8 * (seed % 4), 3 + seed % 6
    out = seed % 4
    gs = bendy.Scene.from_json(txt); cam = gs.find_by_tag("camera"); gs.set_camera_aspect(cam, w / h)
    if seed % 2 == 0:
        gs.set_tuning(slices=1 << (seed % 6), phase_vote=seed % 9)     # every other scene with a pinned block size and vote wait
    buf = bendy.Buffer.new(w, h)
    bendy.Tracer.with_config(bendy.Config(output=bendy.Output(out))).render(gs, cam, bendy.RenderConfig.with_samples(spp), buf, seed=seed)
    torch.cuda.synchronize()
    osc = oracle.Scene(json.loads(txt)); ocam = osc.find_by_tag("camera"); osc.set_camera_aspect(ocam, w / h)
    it, _, seg = oracle.render(osc, ocam, oracle.default_config(samples=spp, recursive=0, output=out), w, h, seed, nthreads=16)
    ok = gs.last_stats().segments == seg and np.array_equal(buf.numpy(), it, equal_nan=True)
    if not ok:
        bad += 1
        print(f"MISMATCH seed {seed}: segments {gs.last_stats().segments} vs {seg}, differing pixels "
              f"{int((buf.numpy() != it).any(axis=-1).sum())}", flush=True)
    if (seed - first) % 50 == 49:
        print(f"... {seed - first + 1} scenes, {bad} mismatches, {time.time() - t0:.0f} s", flush=True)
print(f"fuzz sweep: seeds {first}..{first + n - 1}: {n - bad} of {n} scenes bit-identical to the oracle (frames and segment counts)")
sys.exit(1 if bad else 0)
