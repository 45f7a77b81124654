"""Parity tests proper: the HIP path, called through the C ABI (libbendy_hip.so), against the
CPU oracle and the committed golden framebuffers.

Bar (BASELINE.json north_star): per-channel |mean pixel difference| <= 1e-4 vs the CPU reference
path.  What is actually asserted is stronger: the GPU sums are BIT-IDENTICAL to the oracle's
iterative form (same numerics contract), and within 1e-4 (observed ~1e-6) of the oracle's
recursive form, which nests the products exactly like the reference (tracer/mod.rs:473-482).
Parity with the Rust binary itself is unpinned (see oracle/bt_oracle.h).
"""
import json
import os

import numpy as np
import pytest

from conftest import GOLDEN, scene_path
from helpers import flat_scene_json, gpu_render, gpu_scene, oracle_render, oracle_scene, unshard_numpy

pytestmark = pytest.mark.gpu
TOL = 1e-4   # north_star tolerance on the mean framebuffer


def mean_diff(gpu_rgba, ref_rgba, rays):
    return float(np.abs(gpu_rgba[..., :3] - ref_rgba[..., :3]).max() / rays)


def test_native_library_is_the_one_running(bendy):
    import torch
    assert torch.cuda.is_available()
    maps = open("/proc/self/maps").read()
    assert "libbendy_hip.so" in maps
    assert "gfx950" in torch.cuda.get_device_properties(0).gcnArchName


# ---- committed golden framebuffers -------------------------------------------------------------
def _golden_cases():
    import importlib.util
    spec = importlib.util.spec_from_file_location("make_golden", os.path.join(GOLDEN, "make_golden.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m.CASES


@pytest.mark.parametrize("case", sorted(_golden_cases()))
def test_gpu_matches_golden(bendy, case):
    name, w, h, spp, n, out = _golden_cases()[case]
    g = np.load(os.path.join(GOLDEN, case + ".npz"))
    buf, stats, st = gpu_render(bendy, name, w, h, spp, n=n, output=out)
    rays = spp * max(1, n * n)
    got = buf.numpy()
    assert st == bendy.Status.InProgress and buf.samples == rays
    assert stats.segments == int(g["segments"]) and stats.samples == w * h * rays
    assert np.array_equal(got, g["iterative"])                     # bit-exact, alpha untouched
    assert mean_diff(got, g["recursive"], rays) <= TOL


# ---- live oracle comparisons: sizes, sub-pixel modes, ragged frames ---------------------------------
@pytest.mark.parametrize("name,w,h,spp,n", [
    ("cornell", 256, 256, 1, 0),        # BASELINE configs[0] (C1)
    ("scene", 320, 180, 8, 0),
    ("scene", 160, 90, 2, 2),           # the CLI's default pattern: Subpixel(2)
    ("scene", 96, 54, 1, 3),
    ("cornell2", 128, 128, 16, 0),
    ("volume", 192, 128, 8, 0),
    ("cloud", 192, 128, 8, 0),
    ("scene", 50, 30, 4, 0),            # not a multiple of the 16-pixel tile
    ("cornell", 17, 3, 4, 0),
    ("volume", 1, 1, 16, 0),
    ("scene", 1, 37, 2, 0),
])
def test_gpu_matches_oracle(bendy, oracle, name, w, h, spp, n):
    buf, stats, _ = gpu_render(bendy, name, w, h, spp, n=n)
    it, seg = oracle_render(oracle, name, w, h, spp, n=n, recursive=0)
    rec, _ = oracle_render(oracle, name, w, h, spp, n=n, recursive=1)
    rays = spp * max(1, n * n)
    got = buf.numpy()
    assert stats.segments == seg
    assert np.array_equal(got, it)
    assert mean_diff(got, rec, rays) <= TOL


def test_c2_cornell2_512x512x16_full_frame(bendy, oracle):
    """BASELINE configs[1] (C2) at full size against the oracle (4.2 M samples)."""
    buf, stats, _ = gpu_render(bendy, "cornell2", 512, 512, 16)
    it, seg = oracle_render(oracle, "cornell2", 512, 512, 16, recursive=0, threads=16)
    assert stats.segments == seg and np.array_equal(buf.numpy(), it)
    rec, _ = oracle_render(oracle, "cornell2", 512, 512, 16, recursive=1, threads=16)
    assert mean_diff(buf.numpy(), rec, 16) <= TOL


@pytest.mark.parametrize("output", [1, 2, 3])
@pytest.mark.parametrize("name,w,h", [("scene", 96, 54), ("volume", 96, 64), ("cornell2", 64, 64)])
def test_aov_outputs(bendy, oracle, name, w, h, output):
    """Output::{Albedo, Normal, Depth} (tracer/mod.rs:306-315): first non-pass-through ColorData."""
    buf, _, _ = gpu_render(bendy, name, w, h, 4, output=output)
    it, _ = oracle_render(oracle, name, w, h, 4, output=output, recursive=0)
    rec, _ = oracle_render(oracle, name, w, h, 4, output=output, recursive=1)
    assert np.array_equal(buf.numpy(), it) and np.array_equal(buf.numpy(), rec)


def test_render_config_output_override(bendy, oracle):
    sc, cam = gpu_scene(bendy, "scene", 64, 36)
    buf = bendy.Buffer.new(64, 36)
    tr = bendy.Tracer.with_config(bendy.Config(output=bendy.Output.Full))
    tr.render(sc, cam, bendy.RenderConfig(samples=2, output=bendy.Output.Normal), buf)     # mod.rs:220
    it, _ = oracle_render(oracle, "scene", 64, 36, 2, output=2)
    assert np.array_equal(buf.numpy(), it)


# ---- buffer semantics ----------------------------------------------------------------------------------
def test_host_buffer_path_equals_device_path(bendy):
    dev, _, _ = gpu_render(bendy, "scene", 80, 45, 4)
    host, _, _ = gpu_render(bendy, "scene", 80, 45, 4, device="cpu")
    assert np.array_equal(dev.numpy(), host.numpy()) and host.samples == 4


def test_progressive_calls_accumulate_like_the_reference(bendy, oracle):
    """main.rs:245-254: one sample per call into the same Buffer; `+=` keeps the running sums
    and Buffer::samples grows by samples * n^2 (mod.rs:199)."""
    w, h = 64, 36
    sc, cam = gpu_scene(bendy, "scene", w, h)
    tr = bendy.Tracer.with_config(bendy.Config(chunks_x=8, chunks_y=4))
    buf = bendy.Buffer.new(w, h)
    for i in range(4):
        assert tr.render(sc, cam, bendy.RenderConfig.with_samples_subsample(1, bendy.Subsample(2)), buf) == bendy.Status.InProgress
        assert buf.samples == 4 * (i + 1)
    one, _, _ = gpu_render(bendy, "scene", w, h, 4, n=2)
    assert np.array_equal(buf.numpy(), one.numpy())
    it, _ = oracle_render(oracle, "scene", w, h, 4, n=2)
    assert np.array_equal(buf.numpy(), it)
    assert tr.render(sc, cam, bendy.RenderConfig.with_samples(0), buf) == bendy.Status.Done and buf.samples == 16
    buf.clear()
    assert buf.samples == 0 and float(buf.numpy()[..., :3].sum()) == 0.0 and (buf.numpy()[..., 3] == 1).all()


@pytest.mark.parametrize("name", ["scene", "cornell2"])
def test_progressive_packed_calls_equal_one_deep_call(bendy, oracle, name):
    """The interactive loop on a frame large enough for packed launches: four calls of 1 sample x Subpixel(2), each
    a packed launch adding into the running sums, equal one call of 4 samples (a different launch shape) and the oracle; a scratch
    cap that splits the render into several launches keeps it unpacked, same bits."""
    w, h = 512, 320                                  # 640 tiles x 256 pixels x 4 rays: 1.4 work items per lane of an MI355X
    it, seg = oracle_render(oracle, name, w, h, 4, n=2, threads=16)
    sc, cam = gpu_scene(bendy, name, w, h)
    tr = bendy.Tracer.with_config(bendy.Config(chunks_x=8, chunks_y=4))
    buf = bendy.Buffer.new(w, h)
    segments = 0
    for i in range(4):
        tr.render(sc, cam, bendy.RenderConfig.with_samples_subsample(1, bendy.Subsample(2)), buf)
        st = sc.last_stats()
        assert st.packed >= 1
        segments += st.segments
    assert buf.samples == 16 and segments == seg and np.array_equal(buf.numpy(), it)
    capped, st, _ = gpu_render(bendy, name, w, h, 4, n=2, tuning={"packed": 2, "scratch_cap_bytes": 32 * 20 * 256 * 12 * 8})
    assert st.launches == 2 and st.packed == 0 and np.array_equal(capped.numpy(), it)


def test_prefilled_buffer_is_added_to_not_overwritten(bendy, oracle):
    import torch
    w, h = 48, 27
    sc, cam = gpu_scene(bendy, "scene", w, h)
    buf = bendy.Buffer.new(w, h)
    buf.data[..., :3] = 0.25
    buf.data[..., 3] = 0.5
    bendy.Tracer.new().render(sc, cam, bendy.RenderConfig.with_samples(2), buf)
    torch.cuda.synchronize()
    osc, ocam = oracle_scene(oracle, "scene", w, h)
    pre = np.full((h, w, 4), 0.25, np.float32)
    pre[..., 3] = 0.5
    want, _, _ = oracle.render(osc, ocam, oracle.default_config(samples=2, recursive=0, chunks=(4, 2)), w, h, 0x5EED, rgba=pre)
    assert np.array_equal(buf.numpy(), want)                       # alpha stays 0.5


# ---- config merge quirks (tracer/mod.rs:217-229) -----------------------------------------------------------
def test_q1_max_bounces_override_also_sets_volume_bounces(bendy, oracle):
    buf, stats, _ = gpu_render(bendy, "cloud", 96, 64, 2, max_bounces=3)
    it, seg = oracle_render(oracle, "cloud", 96, 64, 2, max_bounces=3, max_volume_bounces=3)
    assert stats.segments == seg and np.array_equal(buf.numpy(), it)
    # RenderConfig.max_volume_bounces is never read by the reference (mod.rs:224)
    buf2, stats2, _ = gpu_render(bendy, "cloud", 96, 64, 2, max_volume_bounces=2)
    it2, seg2 = oracle_render(oracle, "cloud", 96, 64, 2)
    assert stats2.segments == seg2 and np.array_equal(buf2.numpy(), it2)


def test_volume_step_override(bendy, oracle):
    buf, stats, _ = gpu_render(bendy, "volume", 96, 64, 2, volume_step=0.25)
    it, seg = oracle_render(oracle, "volume", 96, 64, 2, volume_step=0.25)
    assert stats.segments == seg and np.array_equal(buf.numpy(), it)


def test_closed_form_flat_scene_on_gpu(bendy):
    import torch
    color = (0.25, 0.5, 0.75)
    sc = bendy.Scene.from_json(flat_scene_json(sphere_color=color, root_intensity=0.5))
    buf = bendy.Buffer.new(33, 33)
    bendy.Tracer.new().render(sc, sc.find_by_tag("camera"), bendy.RenderConfig.with_samples(4), buf)
    torch.cuda.synchronize()
    img = buf.numpy()
    assert np.array_equal(img[16, 16, :3], np.float32(4) * np.array(color, np.float32))
    assert np.array_equal(img[0, 0, :3], np.float32(4) * np.array([0.5, 0.5, 0.5], np.float32))
    assert sc.last_stats().segments == 33 * 33 * 4


def test_cuboid_and_rect_lights(bendy, oracle, tmp_path):
    """Light sampling of every primitive kind (rect.rs:82-108, cuboid.rs:47-81): turn the short
    box of the Cornell scene into a second (emissive, LIGHT) object."""
    import gzip
    doc = json.loads(gzip.open(scene_path("cornell")).read())
    box = doc["objects"]["collection"]["8"]
    box["flags"]["bits"] = 1
    for face in box["inner"]["Cuboid"]["faces"]:
        face[1]["material"] = 1
    p = tmp_path / "two_lights.json"
    p.write_text(json.dumps(doc))
    w, h, spp = 64, 64, 4
    gs = bendy.Scene.load(p); cam = gs.find_by_tag("camera"); gs.set_camera_aspect(cam, 1.0)
    buf = bendy.Buffer.new(w, h)
    bendy.Tracer.with_config(bendy.Config(chunks_x=8, chunks_y=4)).render(gs, cam, bendy.RenderConfig.with_samples(spp), buf)
    osc = oracle.Scene.load(p); ocam = osc.find_by_tag("camera"); osc.set_camera_aspect(ocam, 1.0)
    assert osc.n_lights() == 2
    it, _, seg = oracle.render(osc, ocam, oracle.default_config(samples=spp, recursive=0), w, h, 0x5EED, nthreads=8)
    assert gs.last_stats().segments == seg and np.array_equal(buf.numpy(), it)


# ---- multi-GPU sharding on one device ---------------------------------------------------------------------
@pytest.mark.parametrize("world", [2, 3, 8])
def test_sharded_render_matches_full_frame(bendy, world):
    """Every rank's shard rendered here in turn; concatenation stands in for the all-gather.
    The image must not depend on the number of ranks (RNG keyed by global pixel / sample)."""
    import torch
    w, h, spp = 200, 120, 4          # 13 x 8 tiles, ragged right/bottom edge, padded last shard
    full, _, _ = gpu_render(bendy, "scene", w, h, spp)
    sc, cam = gpu_scene(bendy, "scene", w, h)
    tr = bendy.Tracer.with_config(bendy.Config(chunks_x=8, chunks_y=4))
    shards = []
    for rank in range(world):
        s = bendy.new_shard(w, h, world)
        assert tr.render_shard(sc, cam, bendy.RenderConfig.with_samples(spp), s, w, h, rank, world) == bendy.Status.InProgress
        shards.append(s)
    gathered = torch.cat(shards)
    out = bendy.Buffer.new(w, h)
    out.data.zero_()
    bendy.unshard(gathered, out, world)
    torch.cuda.synchronize()
    assert np.array_equal(out.numpy(), full.numpy())
    assert np.array_equal(unshard_numpy(gathered.cpu().numpy(), w, h, world), full.numpy())


# ---- sample slicing: few pixels x many samples per pixel (a rank's shard under weak scaling) ---------------------
@pytest.mark.parametrize("name,w,h,spp,n,output", [
    ("scene", 64, 48, 128, 0, 0),
    ("volume", 48, 32, 96, 0, 0),
    ("cornell2", 40, 24, 17, 2, 0),      # 68 rays per pixel, ragged frame, slice bounds not a multiple of n^2
    ("scene", 64, 48, 64, 0, 3),         # AOV through the parked samples
    ("cloud", 32, 32, 1000, 0, 0),
    ("scene", 200, 120, 9, 0, 0),        # 9 samples over 2 slices, ragged right / bottom tiles
])
def test_sliced_render_matches_oracle(bendy, oracle, name, w, h, spp, n, output):
    """bt_api.cpp slices the samples of a pixel over the lanes of a workgroup (BtLaunch::slices), parks every
    sample's value and the last wave of the workgroup sums them in sample order: same bits."""
    buf, stats, _ = gpu_render(bendy, name, w, h, spp, n=n, output=output)
    assert stats.slices > 1
    it, seg = oracle_render(oracle, name, w, h, spp, n=n, output=output, recursive=0)
    assert stats.segments == seg
    assert np.array_equal(buf.numpy(), it)


@pytest.mark.parametrize("slices", [1, 2, 4, 8, 16, 32])
def test_every_slice_count_gives_the_same_frame(bendy, oracle, slices):
    """bt_tuning.slices forces S: each block shape (16x16 ... 4x2 pixels) must give the oracle's bits, on a ragged
    frame, in the full-frame and in the sharded layout."""
    import torch
    w, h, spp, world = 70, 41, 24, 3
    buf, stats, _ = gpu_render(bendy, "cornell", w, h, spp, tuning={"slices": slices})
    assert stats.slices == slices
    it, seg = oracle_render(oracle, "cornell", w, h, spp)
    assert stats.segments == seg and np.array_equal(buf.numpy(), it)
    sc, cam = gpu_scene(bendy, "cornell", w, h, tuning={"slices": slices})
    tr = bendy.Tracer.with_config(bendy.Config(chunks_x=8, chunks_y=4))
    shards = [_render_shard(bendy, tr, sc, cam, w, h, spp, r, world) for r in range(world)]
    out = bendy.Buffer.new(w, h)
    bendy.unshard(torch.cat(shards), out, world)
    torch.cuda.synchronize()
    assert np.array_equal(out.numpy(), it)


@pytest.mark.parametrize("world", [1, 3])
@pytest.mark.parametrize("name,samples,n", [("cornell2", 1, 2), ("scene", 1, 0), ("volume", 3, 0)])
def test_shallow_launches(bendy, oracle, name, samples, n, world):
    """The reference's interactive pattern (main.rs:245-254: one Tracer::render of 1 sample x Subpixel(2) per displayed frame)
    and one ray per pixel: ragged frame, an odd number of tiles, full-frame and sharded layout."""
    import torch
    w, h = 150, 75                                   # 10 x 5 tiles, ragged right / bottom edge
    it, seg = oracle_render(oracle, name, w, h, samples, n=n)
    sc, cam = gpu_scene(bendy, name, w, h)
    tr = bendy.Tracer.with_config(bendy.Config(chunks_x=8, chunks_y=4))
    rc = bendy.RenderConfig.with_samples_subsample(samples, bendy.Subsample(n))
    if world == 1:
        buf = bendy.Buffer.new(w, h)
        tr.render(sc, cam, rc, buf)
        torch.cuda.synchronize()
        assert sc.last_stats().segments == seg and np.array_equal(buf.numpy(), it)
    else:
        shards = []
        for r in range(world):
            s = bendy.new_shard(w, h, world)
            tr.render_shard(sc, cam, rc, s, w, h, r, world)
            shards.append(s)
        out = bendy.Buffer.new(w, h)
        bendy.unshard(torch.cat(shards), out, world)
        torch.cuda.synchronize()
        assert np.array_equal(out.numpy(), it)


def _workgroup_slots():
    import torch
    return torch.cuda.get_device_properties(0).multi_processor_count * 7      # bt_api.cpp: 7 workgroups per CU for small scene tables


@pytest.mark.parametrize("world", [1, 3])
@pytest.mark.parametrize("name,w,h,samples,n,slices", [
    ("cornell2", 330, 200, 3, 0, 0),     # 3 rays per pixel: rows padded to 4 (holes in the queue), block shape left to the library
    ("scene", 400, 260, 1, 2, 16),       # the reference's interactive pattern, 4 x 4 blocks
    ("volume", 330, 200, 5, 0, 32),      # 5 -> 8 rows
    ("cloud", 512, 300, 1, 0, 4),        # one ray per pixel, whole quadrants
    ("cornell", 330, 200, 2, 3, 8),      # 18 -> 32 rows, ragged right / bottom tiles
])
@pytest.mark.parametrize("packed", [1, 2])
def test_packed_launches(bendy, oracle, name, w, h, samples, n, slices, world, packed):
    """bt_tuning.packed = 1: one workgroup per workgroup slot of the GPU, each owning every k-th pixel block behind ONE queue
    (rows of a block padded to a power of two), all its waves summing the parked values at the end; packed = 2: the drain moves
    the paths in flight between lanes through LDS records -- scheduling only, the oracle's bits in the full-frame and in the
    sharded layout."""
    import torch
    it, seg = oracle_render(oracle, name, w, h, samples, n=n, threads=16)
    tuning = {"packed": packed}
    if slices:
        tuning["slices"] = slices
    sc, cam = gpu_scene(bendy, name, w, h, tuning=tuning)
    tr = bendy.Tracer.with_config(bendy.Config(chunks_x=8, chunks_y=4))
    rc = bendy.RenderConfig.with_samples_subsample(samples, bendy.Subsample(n))
    tiles = -(-w // 16) * -(-h // 16)
    if world == 1:
        buf = bendy.Buffer.new(w, h)
        tr.render(sc, cam, rc, buf)
        torch.cuda.synchronize()
        st = sc.last_stats()
        assert st.segments == seg and np.array_equal(buf.numpy(), it)
    else:
        shards = []
        for r in range(world):
            sh = bendy.new_shard(w, h, world)
            tr.render_shard(sc, cam, rc, sh, w, h, r, world)
            shards.append(sh)
        out = bendy.Buffer.new(w, h)
        bendy.unshard(torch.cat(shards), out, world)
        torch.cuda.synchronize()
        st = sc.last_stats()
        assert np.array_equal(out.numpy(), it)
    blocks = -(-tiles // world) * st.slices
    has_drain_rounds = name in ("cornell", "cornell2")          # the compacting drain is compiled into the rect build only
    assert st.packed == ((packed if has_drain_rounds else 1) if blocks > _workgroup_slots() else 0)
    if st.packed:
        assert st.workgroups == _workgroup_slots() and (not slices or st.slices == slices)


def test_packing_is_automatic_for_mid_sized_launches_and_absent_elsewhere(bendy, oracle):
    """bt_api.cpp packs launches of ~1 ... 24 work items per lane of the GPU (the interactive pattern on a 768 x 512 frame:
    4 rays per pixel, 3.4 items per lane on an MI355X), in every Output mode; the lens extension has no packed builds."""
    import torch
    w, h = 768, 512
    it, seg = oracle_render(oracle, "scene", w, h, 1, n=2, threads=16)
    for packed, expect in ((-1, 1), (0, 0), (1, 1), (2, 1)):         # sphere builds have no drain rounds: 2 falls back to 1
        sc, cam = gpu_scene(bendy, "scene", w, h, tuning={"packed": packed})
        buf = bendy.Buffer.new(w, h)
        tr = bendy.Tracer.with_config(bendy.Config(chunks_x=8, chunks_y=4))
        tr.render(sc, cam, bendy.RenderConfig.with_samples_subsample(1, bendy.Subsample(2)), buf)
        torch.cuda.synchronize()
        st = sc.last_stats()
        assert st.packed == expect and st.segments == seg and np.array_equal(buf.numpy(), it)
    # BASELINE configs[1] (C2) is such a launch on the rect build: packed, with the compacting drain
    buf, st, _ = gpu_render(bendy, "cornell2", 512, 512, 16)
    it2, seg2 = oracle_render(oracle, "cornell2", 512, 512, 16, threads=16)
    assert st.packed == 2 and st.segments == seg2 and np.array_equal(buf.numpy(), it2)
    # a deep launch is never packed on its own
    _, st, _ = gpu_render(bendy, "scene", 320, 200, 300)
    assert st.packed == 0
    # the lens extension has no packed builds: asked for, not available -> one block per workgroup (lens tests compare the pixels)
    sc, cam = gpu_scene(bendy, "scene", 400, 260, tuning={"packed": 1})
    sc.set_lens((0.6, 0.4, 4.0), 0.15, 0.1, 6.0, 800)
    buf = bendy.Buffer.new(400, 260)
    bendy.Tracer.with_config(bendy.Config(chunks_x=8, chunks_y=4)).render(sc, cam, bendy.RenderConfig.with_samples(2), buf)
    torch.cuda.synchronize()
    assert sc.last_stats().packed == 0
    # every Output mode has packed builds (the AOV ones without the compacting drain: a path record carries no first-hit state)
    for name, output in (("scene", 2), ("cornell2", 1), ("volume", 3)):
        buf, st, _ = gpu_render(bendy, name, 400, 260, 2, output=output, tuning={"packed": 2})
        it2, seg2 = oracle_render(oracle, name, 400, 260, 2, output=output, threads=16)
        assert st.packed == 1 and st.segments == seg2 and np.array_equal(buf.numpy(), it2)


@pytest.mark.parametrize("max_wait", [0, 1, 2, 7])
@pytest.mark.parametrize("name,w,h,spp", [("scene", 96, 54, 24), ("cloud", 64, 48, 12)])
def test_phase_vote_is_scheduling_only(bendy, oracle, name, w, h, spp, max_wait):
    """Sphere-only builds vote every iteration between the camera event and the scatter / volume events; the losing
    lanes keep their state for at most bt_tuning.phase_vote iterations (0 = no vote).  Same operations per lane, same bits."""
    buf, stats, _ = gpu_render(bendy, name, w, h, spp, tuning={"phase_vote": max_wait})
    it, seg = oracle_render(oracle, name, w, h, spp)
    assert stats.segments == seg and np.array_equal(buf.numpy(), it)


def test_default_sample_base_after_a_change_of_subsample(bendy, oracle):
    """Tracer.render's default sample_base is the next UNUSED sample index (ceil(buffer.samples / n^2)): three calls with
    Subsample::None and then one with Subpixel(2) -- through the default -- must equal the same four calls with the sample bases
    spelled out (0, 1, 2 and then 1 in Subpixel(2)'s numbering, not floor(3 / 4) = 0, which would replay index 0..3), and the
    oracle given those bases."""
    w, h = 64, 40
    sc, cam = gpu_scene(bendy, "scene", w, h)
    tr = bendy.Tracer.with_config(bendy.Config(chunks_x=8, chunks_y=4))
    one, sub = bendy.RenderConfig.with_samples(1), bendy.RenderConfig.with_samples_subsample(1, bendy.Subsample(2))
    a, b_ = bendy.Buffer.new(w, h), bendy.Buffer.new(w, h)
    for _ in range(3):
        tr.render(sc, cam, one, a)
    tr.render(sc, cam, sub, a)
    for i in range(3):
        tr.render(sc, cam, one, b_, sample_base=i)
    tr.render(sc, cam, sub, b_, sample_base=1)
    assert a.samples == b_.samples == 7 and np.array_equal(a.numpy(), b_.numpy())
    first, _ = oracle_render(oracle, "scene", w, h, 3)
    last, _ = oracle_render(oracle, "scene", w, h, 1, n=2, sample_base=1)
    want = first.copy()
    want[..., :3] = first[..., :3] + last[..., :3]          # `*r += pixel.r` call after call (buffer.rs:159-164)
    floor_version = bendy.Buffer.new(w, h)
    for i in range(3):
        tr.render(sc, cam, one, floor_version, sample_base=i)
    tr.render(sc, cam, sub, floor_version, sample_base=0)
    assert not np.array_equal(a.numpy(), floor_version.numpy())
    # (the two oracle frames are summed in float here, the kernel adds sample by sample: equal to a few ulp, not bit for bit)
    assert np.abs(a.numpy()[..., :3] - want[..., :3]).max() <= 1e-5 * max(1.0, float(np.abs(want[..., :3]).max()))


def test_one_deep_call_equals_many_shallow_calls(bendy):
    """Size-independent property (main.rs:245-254 accumulates call after call): 2048 samples in one call -- pixel blocks
    of 4x2, hundreds of items per lane -- give the bits of 32 calls of 64 samples."""
    w, h = 320, 180
    deep, st, _ = gpu_render(bendy, "scene", w, h, 2048)
    assert st.slices == 32
    sc, cam = gpu_scene(bendy, "scene", w, h)
    tr = bendy.Tracer.with_config(bendy.Config(chunks_x=8, chunks_y=4))
    buf = bendy.Buffer.new(w, h)
    for _ in range(32):
        tr.render(sc, cam, bendy.RenderConfig.with_samples(64), buf)
    assert buf.samples == 2048 and np.array_equal(buf.numpy(), deep.numpy())


def test_scratch_is_kept_between_deep_and_shallow_renders_and_trimmed_on_request(bendy, oracle):
    """The parked sample values live in scratch memory on the handle (bt_stats.scratch_bytes).  A caller that alternates deep
    renders with shallow previews keeps it -- it shrinks only after eight shallow renders in a row -- and bt_scene_trim()
    returns it at once; the pixels never depend on it."""
    import torch
    w, h = 160, 96
    sc, cam = gpu_scene(bendy, "scene", w, h)
    tr = bendy.Tracer.with_config(bendy.Config(chunks_x=8, chunks_y=4))
    buf = bendy.Buffer.new(w, h)
    deep, shallow = bendy.RenderConfig.with_samples(64), bendy.RenderConfig.with_samples(2)
    tr.render(sc, cam, deep, buf, sample_base=0)
    held = sc.last_stats().scratch_bytes
    assert held >= w * h * 64 * 12
    for i in range(3):                                         # deep / shallow alternation: nothing is freed
        tr.render(sc, cam, shallow, buf, sample_base=64 + 66 * i)
        assert sc.last_stats().scratch_bytes == held
        tr.render(sc, cam, deep, buf, sample_base=66 + 66 * i)
        assert sc.last_stats().scratch_bytes == held
    base = 64 + 66 * 3
    sizes = []
    for i in range(9):                                         # shallow renders only: the eighth in a row shrinks it
        tr.render(sc, cam, shallow, buf, sample_base=base + 2 * i)
        sizes.append(sc.last_stats().scratch_bytes)
    assert sizes[:7] == [held] * 7 and sizes[7] < held // 4 and sizes[8] == sizes[7]
    tr.render(sc, cam, deep, buf, sample_base=base + 18)
    assert sc.last_stats().scratch_bytes == held
    sc.trim()
    tr.render(sc, cam, shallow, buf, sample_base=base + 18 + 64)
    assert sc.last_stats().scratch_bytes == sizes[7]
    torch.cuda.synchronize()
    total = base + 18 + 64 + 2
    it, _ = oracle_render(oracle, "scene", w, h, total)
    assert buf.samples == total and np.array_equal(buf.numpy(), it)


def test_render_deeper_than_the_scratch_is_split_into_launches(bendy, oracle):
    """A render whose parked samples would not fit the scratch cap is issued as several launches over consecutive
    sample ranges (bt_api.cpp); bt_tuning.scratch_cap_bytes shrinks the cap so that 40 samples need 4 launches
    (12+12+12+4), pinned to the block queue."""
    w, h, spp = 64, 48, 40
    buf, stats, _ = gpu_render(bendy, "volume", w, h, spp, tuning={"scratch_cap_bytes": 64 * 48 * 12 * 12})
    it, seg = oracle_render(oracle, "volume", w, h, spp)
    assert stats.launches == 4 and 0 < stats.scratch_bytes <= 64 * 48 * 12 * 12        # 12 B per parked sample
    assert stats.slices > 1 and stats.segments == seg and stats.samples == w * h * spp
    assert np.array_equal(buf.numpy(), it)


def test_sliced_render_adds_to_prefilled_buffer(bendy, oracle):
    w, h = 48, 32
    sc, cam = gpu_scene(bendy, "scene", w, h)
    tr = bendy.Tracer.with_config(bendy.Config(chunks_x=8, chunks_y=4))
    buf = bendy.Buffer.new(w, h)
    tr.render(sc, cam, bendy.RenderConfig.with_samples(8), buf)               # unsliced
    tr.render(sc, cam, bendy.RenderConfig.with_samples(120), buf)             # sliced, sample_base 8
    assert sc.last_stats().slices > 1 and buf.samples == 128
    it, _ = oracle_render(oracle, "scene", w, h, 128)
    assert np.array_equal(buf.numpy(), it)


def test_sliced_shard_matches_full_frame(bendy):
    import torch
    w, h, spp, world = 200, 120, 96, 3
    full, st, _ = gpu_render(bendy, "scene", w, h, spp)
    sc, cam = gpu_scene(bendy, "scene", w, h)
    tr = bendy.Tracer.with_config(bendy.Config(chunks_x=8, chunks_y=4))
    shards = [_render_shard(bendy, tr, sc, cam, w, h, spp, r, world) for r in range(world)]
    assert sc.last_stats().slices > 1
    out = bendy.Buffer.new(w, h)
    bendy.unshard(torch.cat(shards), out, world)
    torch.cuda.synchronize()
    assert np.array_equal(out.numpy(), full.numpy())


# ---- full BASELINE sizes: size-independent properties + oracle spot checks ---------------------------------
def _host_threads():
    try:
        return max(1, min(64, len(os.sched_getaffinity(0))))
    except AttributeError:
        return max(1, min(64, os.cpu_count() or 1))


def _full_frame(bendy, oracle, name, w, h, spp, seed=0x5EED):
    """Tracer::render end to end (mod.rs:179-202) at a BASELINE.json size: the WHOLE frame of running sums and the number
    of path segments must equal the oracle's (iterative form, every host thread), bit for bit."""
    buf, stats, _ = gpu_render(bendy, name, w, h, spp, seed=seed)
    img = buf.numpy()
    assert stats.samples == w * h * spp and (img[..., 3] == 1.0).all() and np.isfinite(img).all()
    it, seg = oracle_render(oracle, name, w, h, spp, recursive=0, seed=seed, threads=_host_threads())
    assert stats.segments == seg
    assert np.array_equal(img, it), f"{int((img != it).any(axis=-1).sum())} of {w * h} pixels differ"
    return img, it, stats


def test_c3_scene_1080p_64spp(bendy, oracle):
    """BASELINE configs[2] (flat space: the reference has no lens code): full frame + segment count vs the oracle."""
    img, _, stats = _full_frame(bendy, oracle, "scene", 1920, 1080, 64)
    again, stats2, _ = gpu_render(bendy, "scene", 1920, 1080, 64)
    assert np.array_equal(img, again.numpy()) and stats.segments == stats2.segments       # deterministic
    m = img[..., :3].mean() / 64
    lo, _ = oracle_render(oracle, "scene", 192, 108, 16, seed=77, threads=16)
    assert abs(m - lo[..., :3].mean() / 16) / m < 0.03             # independent seed / resolution: same estimator
    # tolerance of BASELINE.json's metric ("pixels within 1e-4 of CPU reference"), against the oracle's recursive form
    # (products nested as mod.rs:473-482 nests them), on the mean framebuffer
    rec, _ = oracle_render(oracle, "scene", 1920, 1080, 64, recursive=1, threads=_host_threads())
    assert np.abs(img[..., :3] - rec[..., :3]).max() / 64 <= 1e-4


def test_c4_volume_1080p_64spp(bendy, oracle):
    """BASELINE configs[3] (volumetric march, flat space): full frame + segment count vs the oracle."""
    _full_frame(bendy, oracle, "volume", 1920, 1080, 64)


def test_c5_scene_4k_256spp_sharded_equals_full(bendy, oracle):
    """BASELINE configs[4] on one GPU: the full 3840x2160x256spp frame equals the oracle's (every pixel, segment count),
    and all eight shards rendered in turn and un-permuted equal that frame."""
    import torch
    w, h, spp = 3840, 2160, 256
    img, _, _ = _full_frame(bendy, oracle, "scene", w, h, spp)
    sc, cam = gpu_scene(bendy, "scene", w, h)
    tr = bendy.Tracer.with_config(bendy.Config(chunks_x=8, chunks_y=4))
    gathered = torch.cat([_render_shard(bendy, tr, sc, cam, w, h, spp, r, 8) for r in range(8)])
    out = bendy.Buffer.new(w, h)
    bendy.unshard(gathered, out, 8)
    torch.cuda.synchronize()
    assert np.array_equal(out.numpy(), img)


def _render_shard(bendy, tr, sc, cam, w, h, spp, rank, world):
    s = bendy.new_shard(w, h, world)
    tr.render_shard(sc, cam, bendy.RenderConfig.with_samples(spp), s, w, h, rank, world)
    return s


def test_statistical_agreement_with_independent_seeds(bendy, oracle):
    """Estimator-level check: GPU and oracle at different seeds agree within Monte-Carlo error."""
    for name, w, h in [("cornell", 64, 64), ("volume", 96, 64)]:
        g, _, _ = gpu_render(bendy, name, w, h, 256, seed=1234)
        c, _ = oracle_render(oracle, name, w, h, 64, seed=4321, threads=16)
        mg, mc = g.numpy()[..., :3].mean(axis=(0, 1)) / 256, c[..., :3].mean(axis=(0, 1)) / 64
        assert np.all(np.abs(mg - mc) / mc < 0.04), (name, mg, mc)


# ---- resolve (next row f-2) and error paths --------------------------------------------------------------------
def test_preview_matches_oracle_resolve(bendy, oracle):
    buf, _, _ = gpu_render(bendy, "scene", 96, 54, 8)
    for cs in (bendy.ColorSpace.Linear, bendy.ColorSpace.SRgb, bendy.ColorSpace.Normal):
        buf.color_space = cs
        got = buf.preview()
        want = oracle.preview(buf.numpy(), buf.samples, int(cs))
        d = np.abs(got.astype(np.int32) - want.astype(np.int32)).max()
        assert d == 0          # numerics contract N9: own exp2/log2 -> the sRGB transfer is bit-exact too
        assert (got[..., 3] == 255).all()


def test_render_errors(bendy):
    sc, cam = gpu_scene(bendy, "scene", 16, 16)
    buf = bendy.Buffer.new(16, 16)
    sphere_ref = 2
    with pytest.raises(bendy.BendyError) as e:
        bendy.Tracer.new().render(sc, sphere_ref, bendy.RenderConfig.with_samples(1), buf)
    assert e.value.code == -5 and buf.samples == 0                  # "expected a camera object" (mod.rs:246)
    with pytest.raises(bendy.BendyError) as e:
        bendy.Tracer.new().render(sc, 12345, bendy.RenderConfig.with_samples(1), buf)
    assert e.value.code == -4


# ---- fuzz: random scenes (every primitive, material and light kind; scaled transforms) -------------------
@pytest.mark.parametrize("seed", range(24))
def test_random_scenes_bit_exact(bendy, oracle, seed, tuning=None):
    import torch
    from scene_gen import random_scene
    txt = random_scene(seed, n_objects=4 + seed % 9)
    w, h, spp = 72, 48, 4
    out = seed % 4 if seed >= 16 else 0
    gs = bendy.Scene.from_json(txt); cam = gs.find_by_tag("camera"); gs.set_camera_aspect(cam, w / h)
    if tuning:
        gs.set_tuning(**tuning)
    buf = bendy.Buffer.new(w, h)
    bendy.Tracer.with_config(bendy.Config(output=bendy.Output(out))).render(gs, cam, bendy.RenderConfig.with_samples(spp), buf, seed=seed)
    torch.cuda.synchronize()
    osc = oracle.Scene(json.loads(txt)); ocam = osc.find_by_tag("camera"); osc.set_camera_aspect(ocam, w / h)
    it, _, seg = oracle.render(osc, ocam, oracle.default_config(samples=spp, recursive=0, output=out), w, h, seed, nthreads=8)
    got = buf.numpy()
    assert gs.last_stats().segments == seg
    assert np.array_equal(got, it, equal_nan=True)


def _compare_json_scene(bendy, oracle, txt, w, h, spp, seed=3):
    import torch
    gs = bendy.Scene.from_json(txt); cam = gs.find_by_tag("camera"); gs.set_camera_aspect(cam, w / h)
    buf = bendy.Buffer.new(w, h)
    bendy.Tracer.new().render(gs, cam, bendy.RenderConfig.with_samples(spp), buf, seed=seed)
    torch.cuda.synchronize()
    osc = oracle.Scene(json.loads(txt)); ocam = osc.find_by_tag("camera"); osc.set_camera_aspect(ocam, w / h)
    it, _, seg = oracle.render(osc, ocam, oracle.default_config(samples=spp, recursive=0), w, h, seed, nthreads=8)
    assert gs.last_stats().segments == seg
    assert np.array_equal(buf.numpy(), it, equal_nan=True)
    return gs


def test_scene_without_primitives(bendy, oracle):
    """Only a camera: every ray goes to sample_root (mod.rs:429-452); empty device tables."""
    doc = json.loads(flat_scene_json())
    del doc["objects"]["collection"]["1"]
    gs = _compare_json_scene(bendy, oracle, json.dumps(doc), 40, 24, 3)
    assert gs.export_prims().shape[0] == 0


def test_density_map_larger_than_the_lds_budget(bendy, oracle):
    """A 24^3 density map (55 KB) stays in global memory instead of LDS; same pixels."""
    from scene_gen import random_scene
    for seed in (100, 101, 102):
        _compare_json_scene(bendy, oracle, random_scene(seed, n_objects=6, volume_prob=1.0, density_dims=(24,)), 64, 40, 4)


# ---- BASELINE configs[1] at full size ---------------------------------------------------------------------------------
def test_c2_cornell2_512_16spp(bendy, oracle):
    """BASELINE configs[1] (cornell2.json.gz 512 x 512 x 16 spp): the whole frame and the segment count against the oracle."""
    _full_frame(bendy, oracle, "cornell2", 512, 512, 16)



def test_volume_fast_paths_and_their_fallbacks(bendy, oracle):
    """Round 2's march shortcuts (BtVolBox + div_refined for the three divisions of Volume::shade, DensityMap::sample
    without the bounds tests) apply to well-formed scenes only; a density map whose `size` exceeds dim - 1 (the reference
    would assert, the oracle returns 0 there) and a volume sphere far outside the reciprocal's range take the exact code.
    All three must give the oracle's bits."""
    import gzip
    doc = json.loads(gzip.open(scene_path("volume")).read())
    vol_key = next(k for k, v in doc["data"]["collection"].items() if "Volume" in v["inner"])
    sph_key = next(k for k, v in doc["objects"]["collection"].items()
                   if isinstance(v["inner"], dict) and "Sphere" in v["inner"] and v["inner"]["Sphere"]["volume"] is not None)
    variants = {"as bundled": doc}
    unsafe = json.loads(json.dumps(doc))
    unsafe["data"]["collection"][vol_key]["inner"]["Volume"]["DensityMap"]["size"] = [9.5, 7.0, 3.0]
    variants["size beyond the map"] = unsafe
    huge = json.loads(json.dumps(doc))
    huge["objects"]["collection"][sph_key]["inner"]["Sphere"]["radius"] = 2.0e6        # bbox size 4e6 > 2^20
    variants["volume sphere outside div_refined's range"] = huge
    for label, d in variants.items():
        _compare_json_scene(bendy, oracle, json.dumps(d), 96, 64, 6)


def test_exact_math_helpers_over_all_inputs():
    """bt_device.hpp computes sqrt, 1/sqrt, the hoisted divisions and the camera's small-angle sin/cos with the core of the
    compiler's own IEEE expansions (sqrt_bt, rsqrt_bt, div_refined, sincos_small_bt).  tools/exact_math_check.hip compares
    each of them on the device with the plain expression it replaces, over all 2^32 bit patterns of the argument
    (2^32 operand pairs for the division, once in a narrow and once in the widest exponent window a caller gates on); `make`
    builds it next to the library."""
    import subprocess
    exe = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bendy_tracer_amd", "exact_math_check")
    assert os.path.exists(exe), "bendy_tracer_amd/exact_math_check is not built (make -C bendy_tracer_amd/csrc)"
    run = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert run.returncode == 0, run.stdout + run.stderr
    lines = [ln for ln in run.stdout.splitlines() if "mismatches" in ln]
    assert len(lines) == 5 and all(ln.rstrip().endswith("mismatches 0") for ln in lines), run.stdout
