"""Known-answer tests that pin the CPU oracle.

The reference ships no golden vectors (its only #[test] asserts nothing, scene/mod.rs:241-261)
and cannot be built or run here, so parity with the Rust binary is UNPINNED.  These KATs are
derived analytically from the reference source plus the
published Random123 vectors for Philox4x32-10.
"""
import gzip
import hashlib
import json
import math

import numpy as np
import pytest

from conftest import scene_path
from helpers import flat_scene_json


# ---- numerics contract building blocks ------------------------------------------------------
def test_philox_random123_vectors(oracle):
    # Random123 kat_vectors, philox4x32 with 10 rounds
    assert oracle.philox([0, 0, 0, 0], [0, 0]) == [0x6627E8D5, 0xE169C58D, 0xBC57AC4C, 0x9B00DBD8]
    assert oracle.philox([0xFFFFFFFF] * 4, [0xFFFFFFFF] * 2) == [0x408F276D, 0x41C83B0E, 0xA20BC7C6, 0x6D5451FD]
    assert oracle.philox([0x243F6A88, 0x85A308D3, 0x13198A2E, 0x03707344], [0xA4093822, 0x299F31D0]) == \
        [0xD16CFE09, 0x94FDCCEB, 0x5001E420, 0x24126EA1]


def test_sincos_accuracy(oracle):
    xs = np.concatenate([np.linspace(-0.6, 6.2832, 4001), [0.0, math.pi / 2, math.pi, 2 * math.pi]]).astype(np.float32)
    worst = 0.0
    for x in xs:
        s, c = oracle.sincos(float(x))
        worst = max(worst, abs(s - math.sin(float(x))), abs(c - math.cos(float(x))))
    assert worst < 2.0e-7
    assert oracle.sincos(0.0) == (0.0, 1.0)


def test_uniform_scale_keeps_samples_in_range(oracle):
    L = oracle.lib()
    max_rand = np.float32(1.0) - np.float32(2.0 ** -23)
    for lo, hi in [(0.0, 1.0), (0.0, 6.2831855), (-0.5, 0.5), (-2.5, 2.5), (-0.0010416667, 0.0010416667)]:
        lo32, hi32 = np.float32(lo), np.float32(hi)
        s_incl = np.float32(L.bto_uniform_scale(lo, hi, 1))
        s_open = np.float32(L.bto_uniform_scale(lo, hi, 0))
        assert s_incl * max_rand + lo32 <= hi32      # new_inclusive: max sample <= high
        assert s_open * max_rand + lo32 < hi32       # new: max sample < high
        assert abs(float(s_incl) - (hi - lo)) < 1e-6 * max(1.0, hi - lo)
    assert np.float32(L.bto_uniform_scale(0.0, 1.0, 1)) == np.float32(1.0) + np.float32(2.0 ** -23)


def test_orthonormal_pair(oracle):
    import ctypes as C
    rng = np.random.default_rng(1)
    for n in list(rng.normal(size=(50, 3))) + [np.array([0, 0, -1.0]), np.array([0, 0, 1.0]), np.array([0, 1.0, 0])]:
        n = (n / np.linalg.norm(n)).astype(np.float32)
        t1, t2 = (C.c_float * 3)(), (C.c_float * 3)()
        oracle.lib().bto_orthonormal_pair((C.c_float * 3)(*n), t1, t2)
        t1, t2 = np.array(t1[:]), np.array(t2[:])
        assert abs(np.dot(t1, t2)) < 1e-6 and abs(np.dot(t1, n)) < 1e-6 and abs(np.dot(t2, n)) < 1e-6
        assert abs(np.linalg.norm(t1) - 1) < 1e-6 and abs(np.linalg.norm(t2) - 1) < 1e-6
    # UnitDisk::new(-Z) frame used for depth of field
    t1, t2 = (C.c_float * 3)(), (C.c_float * 3)()
    oracle.lib().bto_orthonormal_pair((C.c_float * 3)(0, 0, -1), t1, t2)
    assert list(t1) == [1.0, 0.0, 0.0] and list(t2) == [0.0, -1.0, 0.0]


def test_affine_inverse(oracle):
    a = oracle.Affine(oracle.V3(0.9396926, 0.0, -0.34202018), oracle.V3(0, 1, 0), oracle.V3(0.34202018, 0.0, 0.9396926),
                      oracle.V3(-1.2, 1.0, -3.2))
    inv = oracle.Affine()
    import ctypes as C
    oracle.lib().bto_affine_inverse(C.byref(a), C.byref(inv))

    def mat(m):
        return np.array([[m.cx.x, m.cy.x, m.cz.x, m.t.x], [m.cx.y, m.cy.y, m.cz.y, m.t.y],
                         [m.cx.z, m.cy.z, m.cz.z, m.t.z], [0, 0, 0, 1]], dtype=np.float64)
    assert np.allclose(mat(a) @ mat(inv), np.eye(4), atol=1e-6)


# ---- KAT 1: Ray::with_frustum (ray.rs:103-113) ------------------------------------------------
def test_with_frustum(oracle):
    import ctypes as C
    d = (C.c_float * 3)()
    oracle.lib().bto_ray_with_frustum(0.47, 0.47, 0.0, 0.0, d)
    assert list(d) == [0.0, 0.0, -1.0] or (d[0] == 0 and d[1] == 0 and d[2] == -1.0)
    yfov, xfov = 0.2805, 0.4987
    for u, v in [(0.3, -0.2), (-1.0, -1.0), (1.0, 1.0), (0.5, 0.9)]:
        oracle.lib().bto_ray_with_frustum(yfov, xfov, u, v, d)
        yr, xr = -u * xfov / 2, -v * yfov / 2
        want = [-math.cos(xr) * math.sin(yr), math.sin(xr), -math.cos(xr) * math.cos(yr)]
        assert np.allclose(list(d), want, atol=3e-7)
    oracle.lib().bto_ray_with_frustum(yfov, xfov, 0.5, -0.5, d)
    assert d[0] > 0 and d[1] > 0      # u > 0 looks right, v < 0 (top rows) looks up


def test_yxz_composition_reproduces_the_camera_matrix_stored_in_scene_json(oracle):
    """The only reference-held datum with BOTH a Y and an X rotation: scene.json.gz stores its camera's rotation -- written by
    the reference from `Quat::from_euler(YXZ, 10 deg, -5 deg, 0)` -- as the columns of Ry(10 deg) * Rx(-5 deg).  The closed
    form that stands in for `Ray::with_frustum`'s quaternion (ray.rs:103-113; glam's YXZ order is recalled) must
    send (0, 0, -1) where that matrix sends it -- which pins "YXZ = Ry * Rx" and the sign of both angles with data the
    reference itself wrote (cornell2.json.gz == main.rs:107-214 pins the Y rotation alone)."""
    import ctypes as C
    import gzip
    doc = json.loads(gzip.open(scene_path("scene")).read())
    cam = next(o for o in doc["objects"]["collection"].values() if o["tag"] == "camera")
    m = np.array(cam["transform"]["transform_world"][:9], dtype=np.float64).reshape(3, 3).T      # file: columns x, y, z
    a, b = math.radians(10.0), math.radians(-5.0)
    ry = np.array([[math.cos(a), 0, math.sin(a)], [0, 1, 0], [-math.sin(a), 0, math.cos(a)]])
    rx = np.array([[1, 0, 0], [0, math.cos(b), -math.sin(b)], [0, math.sin(b), math.cos(b)]])
    assert np.abs(m - ry @ rx).max() < 2e-7                       # the stored matrix IS Ry(10) Rx(-5), to f32 rounding
    assert np.abs(m - rx @ ry).max() > 1e-2                       # ... and not the other order
    # with_frustum(yfov, xfov, u, v): angles -u * xfov / 2 about Y and -v * yfov / 2 about X; unit fovs: u = -2a, v = -2b
    d = (C.c_float * 3)()
    oracle.lib().bto_ray_with_frustum(1.0, 1.0, -2.0 * a, -2.0 * b, d)
    assert np.abs(np.array(list(d)) - m @ np.array([0.0, 0.0, -1.0])).max() < 3e-7


# ---- KAT 2: Sphere::hit (sphere.rs:121-148, 85-119) ------------------------------------------
def test_sphere_hit(oracle):
    sc = oracle.Scene(json.loads(flat_scene_json()))
    h = oracle.object_hit(sc, 1, (0, 0, 5), (0, 0, -1))
    assert h["face"] == "Front" and h["t"] == 4.0 and list(h["normal"]) == [0, 0, 1] and list(h["position"]) == [0, 0, 1]
    h = oracle.object_hit(sc, 1, (0, 0, 0), (0, 0, -1))
    assert h["face"] == "Back" and h["t"] == 1.0 and list(h["normal"]) == [0, 0, 1]
    assert oracle.object_hit(sc, 1, (0, 2, 5), (0, 0, -1)) is None             # misses
    assert oracle.object_hit(sc, 1, (0, 0, 5), (0, 0, -1), clip=(0.01, 3.9)) is None   # both roots beyond clip.max
    assert oracle.object_hit(sc, 1, (0, 0, 5), (0, 0, -1), clip=(4.5, 1000))["t"] == 6.0  # falls back to far root
    assert oracle.object_hit(sc, 0, (0, 0, 9), (0, 0, -1)) is None             # cameras are never hit


def test_sphere_volume_faces_and_hit_volumetric(oracle):
    sc = oracle.Scene.load(scene_path("volume"))
    vi = next(i for i in range(sc.c.n_objects) if sc._objects[i].volume >= 0)   # c (0, 0.1, 0), r 1
    h = oracle.object_hit(sc, vi, (0, 0.1, 5), (0, 0, -1))
    assert h["face"] == "VolumeFront" and abs(h["t"] - 4.0) < 1e-6
    h = oracle.object_hit(sc, vi, (0, 0.1, 0), (0, 0, -1))
    assert h["face"] == "VolumeBack"
    # hit_volumetric (sphere.rs:150-166): ray.at(clip.max) inside -> Face::Volume at t = clip.max, normal 0
    h = oracle.object_hit(sc, vi, (0, 0.1, 0), (0, 0, -1), clip=(0.0, 0.1), volumetric=True)
    assert h["face"] == "Volume" and h["t"] == np.float32(0.1) and list(h["normal"]) == [0, 0, 0]
    # ... else the ordinary hit finds the exit
    h = oracle.object_hit(sc, vi, (0, 0.1, -0.95), (0, 0, -1), clip=(0.0, 0.1), volumetric=True)
    assert h["face"] == "VolumeBack" and abs(h["t"] - 0.05) < 1e-6


# ---- KAT 3: Rect::hit (rect.rs:110-155) --------------------------------------------------------
def test_rect_hit(oracle):
    sc = oracle.Scene.load(scene_path("cornell"))
    back = sc.object_index[3]        # back wall: T (0, 2.5, -5), z = (0, 0, 1), half extents 2.5
    h = oracle.object_hit(sc, back, (0, 2.5, 0), (0, 0, -1))
    assert h["face"] == "Front" and h["t"] == 5.0 and list(h["normal"]) == [0, 0, 1]        # p < 0 -> Front
    h = oracle.object_hit(sc, back, (0, 2.5, -10), (0, 0, 1))
    assert h["face"] == "Back" and h["t"] == 5.0 and list(h["normal"]) == [0, 0, -1]
    assert oracle.object_hit(sc, back, (3.0, 2.5, 0), (0, 0, -1)) is None      # outside half_width
    assert oracle.object_hit(sc, back, (2.5, 5.0, 0), (0, 0, -1)) is not None  # containment is inclusive (<=)
    assert oracle.object_hit(sc, back, (0, 2.5, 0), (1, 0, 0)) is None         # |d.n| <= 1e-5
    d = np.array([1.0, 0.0, -0.9e-5]); d /= np.linalg.norm(d)
    assert oracle.object_hit(sc, back, (0, 2.5, 0), d) is None                 # still inside the 1e-5 band
    assert oracle.object_hit(sc, back, (0, 2.5, 0), (0, 0, -1), clip=(0.01, 4.9)) is None


# ---- KAT 4: Cuboid::hit keeps the smallest t with strict `<` from clip.max (cuboid.rs:91-102) ---
def test_cuboid_strict_less(oracle):
    sc = oracle.Scene.load(scene_path("cornell"))
    box = sc.object_index[8]         # short box centre (1, 0.6, -1.4), half (0.5, 0.6, 0.5)
    h = oracle.object_hit(sc, box, (1, 0.6, 5), (0, 0, -1))
    # Quirk Q13: Cuboid::new pairs offset -z with Rect::new(x, y) whose z axis is +z (cuboid.rs:19-30), so
    # face normals point INTO the box: a hit from outside is Face::Back with the normal flipped towards the ray.
    assert abs(h["t"] - 5.9) < 1e-5 and h["face"] == "Back" and list(h["normal"]) == [0, 0, 1]
    t = float(h["t"])
    assert oracle.object_hit(sc, box, (1, 0.6, 5), (0, 0, -1), clip=(0.01, t)) is None      # t == clip.max: rejected
    up = np.nextafter(np.float32(t), np.float32(10))
    assert oracle.object_hit(sc, box, (1, 0.6, 5), (0, 0, -1), clip=(0.01, float(up))) is not None
    light = sc.object_index[6]       # a plain Rect accepts t == clip.max (rect.rs:127-129)
    h = oracle.object_hit(sc, light, (0, 0, -2.5), (0, 1, 0))
    assert oracle.object_hit(sc, light, (0, 0, -2.5), (0, 1, 0), clip=(0.01, float(h["t"]))) is not None
    # from inside the box the nearest of the six faces wins
    h = oracle.object_hit(sc, box, (1, 0.6, -1.4), (1, 0, 0))
    assert abs(h["t"] - 0.5) < 1e-6 and h["face"] == "Back" and list(h["normal"]) == [-1, 0, 0]   # +-x, +-y faces: outward
    h = oracle.object_hit(sc, box, (1, 0.6, -1.4), (0, 0, 1))
    assert abs(h["t"] - 0.5) < 1e-6 and h["face"] == "Front" and list(h["normal"]) == [0, 0, -1]  # +-z faces: inward


# ---- light pdfs (sphere.rs:44-61, rect.rs:92-108) -----------------------------------------------
def test_light_pdfs(oracle):
    sc = oracle.Scene.load(scene_path("scene"))
    li = sc.object_index[3]          # sphere_light c (6, 10, 0) r 2
    p = oracle.object_pdf(sc, li, (6, 0, 0), (0, 1, 0))
    assert abs(p - 64.0 / (math.pi * 4.0)) < 1e-5
    assert oracle.object_pdf(sc, li, (6, 0, 0), (1, 0, 0)) is None
    sc = oracle.Scene.load(scene_path("cornell"))
    li = sc.object_index[6]          # light rect at y = 4.999, area 1
    p = oracle.object_pdf(sc, li, (0, 0, -2.5), (0, 1, 0))
    assert abs(p - 4.999 ** 2) < 1e-4
    d = np.array([0.3, 1.0, 0.0]); d /= np.linalg.norm(d)
    assert oracle.object_pdf(sc, li, (0, 0, -2.5), d) is None      # passes beside the 1 x 1 light
    d = np.array([0.05, 1.0, 0.02]); d /= np.linalg.norm(d)
    p = oracle.object_pdf(sc, li, (0, 0, -2.5), d)
    t = 4.999 / d[1]
    assert abs(p - t * t / abs(d[1])) < 1e-3


# ---- KAT 5: reflect / refract / fresnel (math/mod.rs:41-57) -------------------------------------
def test_reflect_refract_fresnel(oracle):
    import ctypes as C
    L = oracle.lib()
    f3 = lambda a: (C.c_float * 3)(*a)
    out = (C.c_float * 3)()
    s = 1 / math.sqrt(2)
    L.bto_reflect(f3((s, -s, 0)), f3((0, 1, 0)), out)
    assert np.allclose(list(out), [s, s, 0], atol=1e-7)
    v = np.array([0.6, -0.8, 0.0], dtype=np.float32)
    L.bto_refract(f3(v), f3((0, 1, 0)), 1.0, out)          # ior 1: straight through
    assert np.allclose(list(out), v, atol=1e-6)
    L.bto_refract(f3((0, -1, 0)), f3((0, 1, 0)), 1 / 1.4, out)
    assert np.allclose(list(out), [0, -1, 0], atol=1e-6)
    for ior in (1.4, 1 / 1.4, 2.0):
        want = ((1 - ior) / (1 + ior)) ** 2
        assert abs(L.bto_fresnel(f3((0, -1, 0)), f3((0, 1, 0)), ior) - want) < 1e-7          # normal incidence
    assert abs(L.bto_fresnel(f3((1, 0, 0)), f3((0, 1, 0)), 1.4) - 1.0) < 1e-6                 # grazing -> 1


# ---- KAT 6: DensityMap (volume.rs:119-167) -------------------------------------------------------
def test_density_map_trilinear(oracle):
    import ctypes as C
    sc = oracle.Scene.load(scene_path("volume"))
    di = next(i for i in range(sc.c.n_data) if sc._data[i].kind == oracle.VOLUME)
    d = sc._data[di]
    buf = sc._density[d.buffer_offset:d.buffer_offset + d.width * d.height * d.depth].reshape(d.depth, d.height, d.width)
    f3 = lambda a: (C.c_float * 3)(*a)
    S = lambda c: oracle.lib().bto_density_sample(C.byref(sc.c), di, f3(c))
    # corners are exact lattice points; index = z*h*w + y*w + x
    for x in (0, 1):
        for y in (0, 1):
            for z in (0, 1):
                assert S((x, y, z)) == buf[z * 7, y * 7, x * 7]
    assert S((-3, 0, 0)) == buf[0, 0, 0] and S((9, 9, 9)) == buf[7, 7, 7]     # clamp to [0, 1]
    # interior lattice points (coordinate k/7 is not exact in f32: tolerance)
    for (x, y, z) in [(2, 3, 4), (5, 1, 6), (3, 3, 3)]:
        assert abs(S((x / 7, y / 7, z / 7)) - buf[z, y, x]) < 1e-5
    # a cell centre is the mean of its 8 corners
    x, y, z = 2, 3, 4
    assert abs(S(((x + .5) / 7, (y + .5) / 7, (z + .5) / 7)) - buf[z:z + 2, y:y + 2, x:x + 2].mean()) < 1e-6


# ---- KAT 7 / 8: Subsample offsets and Buffer::chunks ---------------------------------------------
def test_subsample_offsets(bendy):
    assert list(bendy.Subsample.none()) == [(0.0, 0.0)]
    s = bendy.Subsample.subpixel(2)
    assert list(s) == [(0.0, 0.0), (0.5, 0.0), (0.0, 0.5), (0.5, 0.5)]         # i fastest (mod.rs:96-102)
    assert s.subpixel_count() == 4 and s.subpixel_size() == 0.5
    assert bendy.Subsample.none().subpixel_count() == 1 and bendy.Subsample.none().subpixel_size() == 1.0
    assert len(list(bendy.Subsample.subpixel(3))) == 9


@pytest.mark.parametrize("w,h,cx,cy", [(1920, 1080, 8, 4), (3840, 2160, 8, 4), (10, 7, 3, 2), (5, 5, 8, 4), (256, 256, 4, 2), (1, 1, 4, 2)])
def test_chunks_cover_every_pixel_once(oracle, bendy, w, h, cx, cy):
    b = oracle.chunk_bounds(w, h, cx, cy)
    cover = np.zeros((h, w), dtype=np.int32)
    for x0, y0, x1, y1 in b:
        assert x0 < x1 <= w and y0 < y1 <= h
        cover[y0:y1, x0:x1] += 1
    assert (cover == 1).all()
    # row-major order, ceil-div tile size (buffer.rs:102-115)
    cw, ch = -(-w // cx), -(-h // cy)
    assert tuple(b[0]) == (0, 0, min(cw, w), min(ch, h))
    assert [tuple(int(v) for v in r) for r in b] == bendy.Buffer.new(w, h, device="cpu").chunks(cx, cy)
    if (w, h, cx, cy) == (1920, 1080, 8, 4):
        assert len(b) == 32 and tuple(b[0]) == (0, 0, 240, 270)


# ---- KAT 9: closed-form images ---------------------------------------------------------------------
def test_closed_form_flat_scene(oracle):
    color = (0.25, 0.5, 0.75)
    sc = oracle.Scene(json.loads(flat_scene_json(sphere_color=color, root_intensity=0.5)))
    cam = sc.find_by_tag("camera")
    w = h = 33
    spp = 4
    for rec in (0, 1):
        img, rc, seg = oracle.render(sc, cam, oracle.default_config(samples=spp, recursive=rec), w, h, 7, nthreads=1)
        assert rc == 1 and seg == w * h * spp                    # every path is one segment
        # centre pixels see the Flat sphere: exact constant; corner pixels see the Emissive root
        assert np.array_equal(img[16, 16, :3], np.float32(spp) * np.array(color, np.float32))
        assert np.array_equal(img[0, 0, :3], np.float32(spp) * np.array([0.5, 0.5, 0.5], np.float32))
        assert (img[..., 3] == 1.0).all()                        # alpha untouched (buffer.rs:159-164)
    # AOVs of the same scene
    alb, _, _ = oracle.render(sc, cam, oracle.default_config(samples=1, output=oracle.OUT_ALBEDO), w, h, 7)
    assert np.array_equal(alb[16, 16, :3], np.array(color, np.float32))      # from_emitted: albedo = emitted
    assert np.array_equal(alb[0, 0, :3], np.zeros(3, np.float32))            # Emissive root: default ColorData
    dep, _, _ = oracle.render(sc, cam, oracle.default_config(samples=1, output=oracle.OUT_DEPTH), w, h, 7)
    assert dep[16, 16, 0] == 1.0 and dep[0, 0, 0] == 1.0                     # depth = +inf -> clamps to 1
    nrm, _, _ = oracle.render(sc, cam, oracle.default_config(samples=1, output=oracle.OUT_NORMAL), w, h, 7)
    assert np.array_equal(nrm[16, 16, :3], np.zeros(3, np.float32))          # from_emitted: normal 0


def test_samples_zero_is_done(oracle):
    sc = oracle.Scene.load(scene_path("cornell"))
    img, rc, seg = oracle.render(sc, sc.find_by_tag("camera"), oracle.default_config(samples=0), 8, 8, 1)
    assert rc == 0 and seg == 0 and (img[..., :3] == 0).all()               # Status::Done (mod.rs:186-188)


# ---- KAT 10: resolve (buffer.rs:117-138, color.rs:14-24) -----------------------------------------
def test_preview_resolve(oracle):
    rgba = np.zeros((1, 8, 4), dtype=np.float32)
    rgba[..., 3] = 1.0
    vals = [0.0, 0.999, 1.0, 2.0, -1.0, 0.5, 0.0031308, 0.2]
    for i, v in enumerate(vals):
        rgba[0, i, :3] = v * 4          # 4 samples
    lin = oracle.preview(rgba, 4, 2)
    assert list(lin[0, :, 0]) == [0, 254, 255, 255, 0, 127, 0, 51]          # (x * 255) as u8 truncates, saturates
    assert (lin[..., 3] == 255).all()
    srgb = oracle.preview(rgba, 4, 3)
    knee = 12.92 * 0.0031308
    assert srgb[0, 6, 0] == int(knee * 255)
    assert srgb[0, 5, 0] == int((1.055 * 0.5 ** (1 / 2.4) - 0.055) * 255)  # 187
    assert srgb[0, 2, 0] == 254 or srgb[0, 2, 0] == 255


def test_srgb_transfer_tracks_libm(oracle):
    """Numerics contract N9: the oracle's own x^(1/2.4) stays within 1 LSB of the exact transfer."""
    xs = np.linspace(0.0, 1.2, 4096).astype(np.float32)
    rgba = np.zeros((1, xs.size, 4), dtype=np.float32)
    rgba[0, :, 0] = xs
    rgba[..., 3] = 1.0
    got = oracle.preview(rgba, 1, 3)[0, :, 0].astype(np.int32)
    x64 = xs.astype(np.float64)
    exact = np.where(x64 <= 0.0031308, 12.92 * x64, 1.055 * np.power(x64, 1 / 2.4) - 0.055)
    want = np.clip(np.floor(exact * 255.0), 0, 255).astype(np.int32)
    assert np.abs(got - want).max() <= 1 and (got != want).mean() < 0.01


# ---- KAT 11: loader --------------------------------------------------------------
APPENDIX_A = {
    "scene": ("24aae0e59700abf7", 6, 7, 202, 74.8469609, 1, [3]),
    "cornell": ("f15ebade33f21348", 9, 5, 513, 266.02077, 0, [6]),
    "cornell2": ("83a5ac4d849bdeeb", 9, 6, 517, 276.68077, 0, [6]),
    "volume": ("8a3e227407682c5f", 5, 6, 684, 105.185692, 1, [4]),
    "cloud": ("ae6a7fe27b62b6c0", 5, 6, 4268, 297.122336, 1, [4]),
}


def _leaves(v):
    if isinstance(v, bool):
        return
    if isinstance(v, (int, float)):
        yield float(v)
    elif isinstance(v, dict):
        for x in v.values():
            yield from _leaves(x)
    elif isinstance(v, list):
        for x in v:
            yield from _leaves(x)


@pytest.mark.parametrize("name", sorted(APPENDIX_A))
def test_bundled_scene_fixtures(oracle, name):
    sha, n_obj, n_data, n_leaves, leaf_sum, root, lights = APPENDIX_A[name]
    raw = gzip.open(scene_path(name)).read()
    assert hashlib.sha256(raw).hexdigest()[:16] == sha
    doc = json.loads(raw)
    leaves = list(_leaves(doc))
    assert len(leaves) == n_leaves and abs(math.fsum(leaves) - leaf_sum) < 1e-5
    sc = oracle.Scene(doc)
    assert sc.c.n_objects == n_obj and sc.c.n_data == n_data
    assert sc._data[sc.c.root_material].data_ref == root
    assert [sc.object_keys[i] for i in range(n_obj) if sc._objects[i].flags & 1] == lights
    assert sc.find_by_tag("camera") is not None and sc._objects[sc.find_by_tag("camera")].kind == oracle.CAMERA
