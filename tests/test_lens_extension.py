"""the gravitational-lens mode.  NOT IN THE REFERENCE (F1: bendy-tracer v1 traces
straight rays, tracer/ray.rs:115-117) -- an additive, default-off extension with no reference behaviour to
match ("parity unpinned").  It is validated analytically against the Schwarzschild metric and by
GPU == CPU-oracle, and it must leave every pixel untouched when it is not switched on."""
import json
import math

import numpy as np
import pytest

from conftest import scene_path
from helpers import flat_scene_json, oracle_scene

LENS = dict(centre=(0.6, 0.4, 4.0), rs=0.15, step=0.2, radius=6.0, max_steps=400)


def deflection(oracle, b, rs, step=0.25, radius=200.0, start=-400.0):
    cfg = oracle.default_config(lens=dict(centre=(0, 0, 0), rs=rs, step=step, radius=radius, max_steps=200000))
    st, pos, d = oracle.lens_trace_free(cfg, (start, b, 0.0), (1, 0, 0))
    return st, math.atan2(-d[1], d[0])


def test_weak_field_deflection_is_4gm_over_c2b(oracle):
    """alpha = 2 rs / b (= 4GM / c^2 b) to first order; second order: (1 + 15 pi / 32 * rs / b)."""
    rs = 0.01
    for b in (1.0, 2.0, 5.0):
        st, alpha = deflection(oracle, b, rs)
        assert st == 0
        first = 2 * rs / b
        assert abs(alpha / first - (1 + 15 * math.pi / 32 * rs / b)) < 2e-3
    # halving the step changes nothing at this accuracy (RK4 is converged)
    assert abs(deflection(oracle, 1.0, rs, step=0.5)[1] - deflection(oracle, 1.0, rs, step=0.1)[1]) < 1e-6


def test_photon_sphere_capture_threshold(oracle):
    """Rays with impact parameter below b_c = 3 sqrt(3) / 2 rs fall into the horizon, above it they escape."""
    rs = 1.0
    bc = 1.5 * math.sqrt(3) * rs

    def status(b):
        cfg = oracle.default_config(lens=dict(centre=(0, 0, 0), rs=rs, step=0.02, radius=60.0, max_steps=400000))
        return oracle.lens_trace_free(cfg, (-100.0, b, 0.0), (1, 0, 0))[0]
    assert [status(bc * f) for f in (0.5, 0.9, 0.995)] == [1, 1, 1]
    assert [status(bc * f) for f in (1.005, 1.1, 2.0)] == [0, 0, 0]


def test_zero_mass_lens_is_a_straight_ray(oracle):
    cfg = oracle.default_config(lens=dict(centre=(0, 0, 0), rs=0.0, step=0.25, radius=10.0, max_steps=10000))
    st, pos, d = oracle.lens_trace_free(cfg, (-30.0, 1.0, 0.5), (1, 0.02, 0.01))
    want = np.array([1, 0.02, 0.01]) / np.linalg.norm([1, 0.02, 0.01])
    assert st == 0 and np.allclose(d, want, atol=2e-6)


def test_lens_off_is_the_default_and_changes_nothing(oracle):
    sc, cam = oracle_scene(oracle, "scene", 48, 27)
    a, _, sa = oracle.render(sc, cam, oracle.default_config(samples=2), 48, 27, 5)
    cfg = oracle.default_config(samples=2, lens=LENS)
    cfg.lens_on = 0                                   # parameters present, switch off
    b, _, sb = oracle.render(sc, cam, cfg, 48, 27, 5)
    assert sa == sb and np.array_equal(a, b)


def test_massless_lens_image_converges_to_the_flat_image(oracle):
    """rs = 0: chords are straight, so the closed-form Flat scene is reproduced except at silhouette pixels
    (hits are found chord by chord, so t differs in the last bits)."""
    sc = oracle.Scene(json.loads(flat_scene_json()))
    cam = sc.find_by_tag("camera")
    flat, _, _ = oracle.render(sc, cam, oracle.default_config(samples=4), 33, 33, 7)
    lens = dict(centre=(0.0, 0.0, 2.5), rs=0.0, step=0.2, radius=3.0, max_steps=1000)
    bent, _, _ = oracle.render(sc, cam, oracle.default_config(samples=4, lens=lens), 33, 33, 7)
    assert (np.abs(flat - bent).max(axis=-1) > 1e-6).mean() < 0.03
    assert np.array_equal(flat[16, 16], bent[16, 16]) and np.array_equal(flat[0, 0], bent[0, 0])


def test_lens_bends_the_image(oracle):
    """A mass in front of the Flat sphere: the centre pixel still sees the sphere or the hole, the frame changes,
    paths swallowed by the horizon are black."""
    sc = oracle.Scene(json.loads(flat_scene_json(root_intensity=0.5)))
    cam = sc.find_by_tag("camera")
    flat, _, _ = oracle.render(sc, cam, oracle.default_config(samples=2), 48, 48, 3)
    lens = dict(centre=(0.0, 0.0, 2.8), rs=0.12, step=0.1, radius=3.0, max_steps=2000)
    rec, _, s1 = oracle.render(sc, cam, oracle.default_config(samples=2, lens=lens, recursive=1), 48, 48, 3)
    it, _, s2 = oracle.render(sc, cam, oracle.default_config(samples=2, lens=lens, recursive=0), 48, 48, 3)
    assert s1 == s2 and np.abs(rec - it).max() <= 1e-5
    assert not np.array_equal(flat, rec)
    assert (rec[..., :3].sum(axis=-1) == 0).sum() > 0                  # the shadow of the horizon
    # the sphere sits exactly behind the mass: its image is an Einstein ring out at the frame's edge (the corner
    # pixel, plain sky without the lens, now shows the sphere), and the centre is the shadow of the horizon
    sphere, sky = np.float32(2) * np.array([0.25, 0.5, 0.75], np.float32), np.float32(2) * np.float32(0.5)
    assert np.array_equal(flat[0, 0, :3], [sky] * 3) and np.array_equal(rec[0, 0, :3], sphere)
    assert np.array_equal(flat[24, 24, :3], sphere) and np.array_equal(rec[24, 24, :3], [0, 0, 0])


# ---------------------------------------------------------------------------------------------- GPU
def _gpu_lens_render(bendy, name, w, h, spp, lens, output=0, seed=11):
    import torch
    sc = bendy.Scene.load(scene_path(name)); cam = sc.find_by_tag("camera"); sc.set_camera_aspect(cam, w / h)
    if lens:
        sc.set_lens(**lens)
    buf = bendy.Buffer.new(w, h)
    bendy.Tracer.with_config(bendy.Config(output=bendy.Output(output))).render(sc, cam, bendy.RenderConfig.with_samples(spp), buf, seed=seed)
    torch.cuda.synchronize()
    return buf.numpy(), sc.last_stats(), sc


@pytest.mark.gpu
@pytest.mark.parametrize("name,w,h,output", [("scene", 96, 54, 0), ("cornell2", 64, 64, 0), ("volume", 72, 48, 0),
                                             ("scene", 64, 36, 3), ("scene", 64, 36, 2), ("cloud", 48, 32, 1)])
def test_gpu_lens_matches_oracle(bendy, oracle, name, w, h, output):
    lens = dict(LENS)
    if name.startswith("cornell"):
        lens.update(centre=(0.3, 2.2, 3.0), rs=0.1)
    got, stats, _ = _gpu_lens_render(bendy, name, w, h, 4, lens, output=output)
    sc, cam = oracle_scene(oracle, name, w, h)
    it, _, seg = oracle.render(sc, cam, oracle.default_config(samples=4, recursive=0, output=output, lens=lens), w, h, 11, nthreads=8)
    assert stats.segments == seg and stats.lens_steps > 0
    assert np.array_equal(got, it, equal_nan=True)
    rec, _, _ = oracle.render(sc, cam, oracle.default_config(samples=4, recursive=1, output=output, lens=lens), w, h, 11, nthreads=8)
    assert np.abs(got[..., :3] - rec[..., :3]).max() / 4 <= 1e-4


@pytest.mark.gpu
def test_gpu_lens_off_restores_the_reference_path(bendy, oracle):
    lens_img, _, sc = _gpu_lens_render(bendy, "scene", 96, 54, 4, LENS)
    plain, st_plain, _ = _gpu_lens_render(bendy, "scene", 96, 54, 4, None)
    assert not np.array_equal(lens_img, plain) and st_plain.lens_steps == 0
    sc.clear_lens()
    import torch
    buf = bendy.Buffer.new(96, 54)
    bendy.Tracer.new().render(sc, sc.find_by_tag("camera"), bendy.RenderConfig.with_samples(4), buf, seed=11)
    torch.cuda.synchronize()
    assert np.array_equal(buf.numpy(), plain)
    with pytest.raises(bendy.BendyError):
        sc.set_lens(centre=(0, 0, 0), rs=1.0, step=0.0, radius=5.0)          # step must be positive
    with pytest.raises(bendy.BendyError):
        sc.set_lens(centre=(0, 0, 0), rs=2.0, step=0.1, radius=1.0)          # radius must exceed rs
