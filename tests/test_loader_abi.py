"""Host logic without a GPU: the C ABI loads and exports every declared symbol, the C++
scene loader (bt_scene.cpp) agrees with the independent Python loader of the oracle, and
error behaviour mirrors the reference's panics / serde errors."""
import ctypes as C
import gzip
import json
import os
import re

import numpy as np
import pytest

from conftest import ROOT, scene_path
from helpers import flat_scene_json

ALL_SCENES = ["scene", "cornell", "cornell2", "volume", "cloud"]


def test_library_exports_every_declared_symbol(bendy):
    header = open(os.path.join(ROOT, "include", "bendy_hip.h")).read()
    header = re.sub(r"/\*.*?\*/", "", header, flags=re.S)
    declared = set(re.findall(r"\b(bt_[a-z_0-9]+)\s*\(", header))
    declared -= {"bt_status", "bt_output"}
    assert len(declared) >= 20
    lib = C.CDLL(bendy.api.LIB_PATH)
    for name in sorted(declared):
        assert hasattr(lib, name), f"{name} declared in include/bendy_hip.h but not exported"
    assert set(bendy.api.EXPORTS) == declared
    assert b"gfx950" in bendy.api.lib.bt_version()


def test_config_defaults_match_reference(bendy):
    c = bendy.api._CConfig()
    bendy.api.lib.bt_config_default(C.byref(c))
    # Config::DEFAULT, tracer/mod.rs:29-38
    assert (c.max_bounces, c.max_volume_bounces, c.chunks_x, c.chunks_y, c.output) == (8, 32, 4, 2, 0)
    assert (c.clip_min, c.clip_max, c.volume_step) == (np.float32(0.01), 1000.0, np.float32(0.1))
    r = bendy.api._CRenderConfig()
    bendy.api.lib.bt_render_config_default(C.byref(r))
    # RenderConfig::DEFAULT, tracer/mod.rs:128-135
    assert r.samples == 64 and r.subsample_n == 0 and not (r.has_output or r.has_max_bounces or r.has_volume_step)
    py = bendy.Config()
    assert (py.max_bounces, py.max_volume_bounces, py.chunks_x, py.chunks_y) == (8, 32, 4, 2)
    assert bendy.RenderConfig().samples == 64 and bendy.RenderConfig.with_samples(3).samples == 3


def _expected_prims(o, sc):
    """Flattened primitive rows recomputed from the oracle's (Python-loaded) scene."""
    f32 = np.float32

    def v(a):
        return np.array([a.x, a.y, a.z], dtype=f32)

    def xf_vector(m, p):
        return (v(m.cx) * p[0] + v(m.cy) * p[1]) + v(m.cz) * p[2]

    rows = []
    mats = [i for i in range(sc.c.n_data) if sc._data[i].kind != o.VOLUME]
    vols = [i for i in range(sc.c.n_data) if sc._data[i].kind == o.VOLUME]

    def unit_axis(a):
        c = [a.x, a.y, a.z]
        ones = [i for i in range(3) if abs(c[i]) == 1.0]
        return ones[0] if len(ones) == 1 and all(c[i] == 0.0 for i in range(3) if i != ones[0]) else -1

    def rect_row(r, tf, oi, strict):
        inv = o.Affine()
        o.lib().bto_affine_inverse(C.byref(tf), C.byref(inv))
        row = np.zeros(36, dtype=f32)
        # shape 1 = general rect, 2 = axis-aligned rect (identity matrix, signed unit axes), 3 = such a rect whose
        # world normal is a signed unit axis too, 4 = signed unit LOCAL axes under any transform; | 8 = cuboid face
        identity = [tf.cx.x, tf.cx.y, tf.cx.z, tf.cy.x, tf.cy.y, tf.cy.z, tf.cz.x, tf.cz.y, tf.cz.z] == [1, 0, 0, 0, 1, 0, 0, 0, 1]
        au, av = unit_axis(r.x), unit_axis(r.y)
        shape = 4 if (au >= 0 and av >= 0 and au != av) else 1     # 4: local axes are signed unit vectors, any transform
        if identity and au >= 0 and av >= 0:
            shape = 2
        nrm = xf_vector(tf, v(r.z))
        aw = unit_axis(o.V3(*[float(x) for x in nrm]))
        if shape == 2 and au != av and aw >= 0 and aw not in (au, av):
            shape = 3
        kind = shape | (8 if strict else 0)
        if shape >= 2:
            row[19:20].view(np.int32)[:] = au
            row[23:24].view(np.int32)[:] = av
        if shape == 3:
            row[27:28].view(np.int32)[:] = aw
        row[:4].view(np.int32)[:] = [kind, oi, mats.index(r.material), -1]
        row[4:7] = xf_vector(tf, v(r.z))
        row[8:11] = v(tf.t)
        row[11] = f32(r.half_width) * f32(r.half_width)
        row[12:15], row[15] = v(inv.cx), f32(r.half_height) * f32(r.half_height)
        row[16:19], row[20:23], row[24:27] = v(inv.cy), v(inv.cz), v(inv.t)
        row[28:31], row[32:35] = v(r.x), v(r.y)
        if shape == 3:       # the six constants of the axis-aligned test side by side instead of Rect.x / Rect.y
            comp = lambda a, i: [a.x, a.y, a.z][i]
            a, b = (1 if aw == 0 else 0), (1 if aw == 2 else 2)
            wq, hq = f32(r.half_width) * f32(r.half_width), f32(r.half_height) * f32(r.half_height)
            row[28:32] = [comp(tf.t, aw), comp(inv.t, a), comp(inv.t, b), wq if au == a else hq]
            row[32:36] = [hq if au == a else wq, float(nrm[aw]), 0.0, 0.0]
        if shape == 4:       # rows u, v of the inverse transform (with its translation) instead of Rect.x / Rect.y
            comp = lambda a, i: [a.x, a.y, a.z][i]
            row[28:32] = [comp(inv.cx, au), comp(inv.cy, au), comp(inv.cz, au), comp(inv.t, au)]
            row[32:36] = [comp(inv.cx, av), comp(inv.cy, av), comp(inv.cz, av), comp(inv.t, av)]
        return row

    for oi in range(sc.c.n_objects):
        ob = sc._objects[oi]
        if ob.kind == o.SPHERE:
            row = np.zeros(36, dtype=f32)
            row[:4].view(np.int32)[:] = [0, oi, mats.index(ob.material), vols.index(ob.volume) if ob.volume >= 0 else -1]
            row[4:7], row[7] = v(ob.world.t), ob.radius
            rows.append(row)
        elif ob.kind == o.RECT:
            rows.append(rect_row(ob.rect, ob.world, oi, False))
        elif ob.kind == o.CUBOID:
            for f in range(6):
                tf = o.Affine(ob.world.cx, ob.world.cy, ob.world.cz, ob.world.t)
                t = xf_vector(ob.world, v(ob.face_offset[f])) + v(ob.world.t)
                tf.t = o.V3(*[float(x) for x in t])
                rows.append(rect_row(ob.faces[f], tf, oi, True))
    return np.array(rows, dtype=f32)


@pytest.mark.parametrize("name", ALL_SCENES)
def test_cpp_loader_matches_python_loader(bendy, oracle, name):
    gs = bendy.Scene.load(scene_path(name))
    os_ = oracle.Scene.load(scene_path(name))
    assert gs.object_count == os_.c.n_objects and gs.data_count == os_.c.n_data
    assert gs.find_by_tag("camera") == os_.object_keys[os_.find_by_tag("camera")]
    assert gs.find_by_tag("no-such-tag") is None
    got = gs.export_prims()
    want = _expected_prims(oracle, os_)
    assert got.shape == want.shape
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))      # bit-exact, including -0.0


def test_from_json_and_plain_json_file(bendy, tmp_path):
    raw = gzip.open(scene_path("scene")).read()
    a = bendy.Scene.from_json(raw)
    p = tmp_path / "scene.json"                  # main.rs:97-102: no .gz extension -> plain JSON
    p.write_bytes(raw)
    b_ = bendy.Scene.load(p)
    c = bendy.Scene.load(scene_path("scene"))
    bits = lambda s: s.export_prims().view(np.uint32)     # int fields (-1) read as NaN floats: compare bits
    assert np.array_equal(bits(a), bits(b_)) and np.array_equal(bits(a), bits(c))


def test_number_forms(bendy):
    """-0.0, exponents (1.8626451e-09) and null appear in the bundled files."""
    doc = json.loads(flat_scene_json())
    txt = json.dumps(doc).replace('"radius": 1.0', '"radius": 1.0e0')
    s = bendy.Scene.from_json(txt)
    assert s.export_prims()[0, 7] == 1.0
    raw = gzip.open(scene_path("volume")).read().decode()
    assert "e-09" in raw or "e-9" in raw
    assert "-0.0" in gzip.open(scene_path("cornell")).read().decode()


def test_error_behaviour(bendy, tmp_path):
    E = bendy.BendyError
    with pytest.raises(E) as e:
        bendy.Scene.load(tmp_path / "missing.json.gz")
    assert e.value.code == -2                                        # io error (main.rs:94)
    with pytest.raises(E) as e:
        bendy.Scene.from_json('{"roots": [], "root_material": 0')
    assert e.value.code == -3                                        # serde error
    with pytest.raises(E):
        bendy.Scene.from_json('{"roots": [], "root_material": 0, "objects": {"collection": {}}}')   # missing field
    bad = tmp_path / "bad.json.gz"
    bad.write_bytes(b"not gzip at all")
    with pytest.raises(E):
        bendy.Scene.load(bad)

    s = bendy.Scene.from_json(flat_scene_json())
    with pytest.raises(E) as e:
        s.set_camera_aspect(1, 1.0)                                  # object 1 is a sphere
    assert e.value.code == -5                                        # "expected a camera object" (mod.rs:246)
    with pytest.raises(E) as e:
        s.set_camera_aspect(99, 1.0)
    assert e.value.code == -4                                        # "invalid object ref" (scene/mod.rs:132)

    # invalid data ref / non-material data behind a material ref (scene/mod.rs:136, mod.rs:464)
    doc = json.loads(flat_scene_json())
    doc["objects"]["collection"]["1"]["inner"]["Sphere"]["material"] = 77
    with pytest.raises(E) as e:
        bendy.Scene.from_json(json.dumps(doc)).export_prims()
    assert e.value.code == -4
    vol = {"inner": {"Volume": {"DensityMap": {"width": 1, "height": 1, "depth": 1, "size": [0, 0, 0], "buffer": [0.5]}}}}
    doc = json.loads(flat_scene_json(extra_data={"3": vol}))
    doc["objects"]["collection"]["1"]["inner"]["Sphere"]["material"] = 3
    with pytest.raises(E) as e:
        bendy.Scene.from_json(json.dumps(doc)).export_prims()
    assert e.value.code == -6
    # Diffuse material but no LIGHT object: the reference panics in Uniform::new(0, 0) (material.rs:112)
    diffuse = {"inner": {"Material": {"Diffuse": {"albedo": {"r": .5, "g": .5, "b": .5}, "roughness": 1.0}}}}
    doc = json.loads(flat_scene_json(extra_data={"3": diffuse}))
    doc["objects"]["collection"]["1"]["inner"]["Sphere"]["material"] = 3
    with pytest.raises(E) as e:
        bendy.Scene.from_json(json.dumps(doc)).export_prims()
    assert e.value.code == -7


def test_render_fails_loudly_without_gpu(bendy):
    """No CPU fallback: on a machine without a HIP device the render entry points return
    BT_ERR_DEVICE instead of producing pixels some other way."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    s = bendy.Scene.load(scene_path("cornell"))
    buf = bendy.Buffer.new(8, 8, device="cpu")
    with pytest.raises(bendy.BendyError) as e:
        bendy.Tracer.new().render(s, s.find_by_tag("camera"), bendy.RenderConfig.with_samples(1), buf)
    assert e.value.code == -8
    assert buf.samples == 0 and (buf.data[..., :3] == 0).all()


def test_samples_zero_is_done_without_touching_the_device(bendy):
    s = bendy.Scene.load(scene_path("cornell"))
    buf = bendy.Buffer.new(8, 8, device="cpu")
    st = bendy.Tracer.new().render(s, s.find_by_tag("camera"), bendy.RenderConfig.with_samples(0), buf)
    assert st == bendy.Status.Done and buf.samples == 0              # mod.rs:186-188


def test_shard_geometry(bendy):
    assert bendy.shard_floats(1920, 1080, 1) == 120 * 68 * 256 * 4
    assert bendy.shard_floats(1920, 1080, 8) == (120 * 68 // 8) * 256 * 4
    assert bendy.shard_floats(100, 50, 3) == -(-(7 * 4) // 3) * 256 * 4      # padded to equal shards
    m = bendy.tile_owner_map(64, 48, 3)
    assert m.shape == (3, 4) and list(m.reshape(-1)) == [i % 3 for i in range(12)]


def test_tuning_is_per_handle_and_validated(bendy, monkeypatch):
    """bt_tuning (include/bendy_hip.h): launch-shape knobs live on the scene handle, not in the environment or in a
    process global; out-of-set values are rejected; NULL restores the defaults."""
    a = bendy.Scene.load(scene_path("scene"))
    b = bendy.Scene.load(scene_path("scene"))
    default = dict(slices=0, phase_vote=-1, scratch_cap_bytes=0, packed=-1, reserved=0)
    assert a.tuning() == default
    a.set_tuning(slices=8, phase_vote=5, scratch_cap_bytes=1 << 20, packed=1)
    assert a.tuning() == {**default, "slices": 8, "phase_vote": 5, "scratch_cap_bytes": 1 << 20, "packed": 1}
    assert b.tuning() == default                     # another handle is untouched
    a.set_tuning(phase_vote=0)                       # fields not named keep their value
    assert a.tuning()["slices"] == 8 and a.tuning()["phase_vote"] == 0
    for bad in (dict(slices=3), dict(slices=64), dict(phase_vote=-2), dict(phase_vote=65), dict(packed=3), dict(packed=-2)):
        with pytest.raises(bendy.BendyError) as e:
            a.set_tuning(**bad)
        assert e.value.code == -1
    a.set_tuning()
    assert a.tuning() == default
    # the library ignores the environment; only the helper for tools / tests translates it
    monkeypatch.setenv("BT_SLICES", "16")
    monkeypatch.setenv("BT_PHASE_VOTE", "2")
    c = bendy.Scene.load(scene_path("scene"))
    assert c.tuning() == default
    assert c.tuning_from_env() == {"slices": 16, "phase_vote": 2} and c.tuning()["slices"] == 16
    with pytest.raises(TypeError):
        c.set_tuning(tiles_per_wg=2)                 # knobs that lost every measurement are gone (so are round 3's queue, end_game and march_pool)
    src = open(os.path.join(ROOT, "bendy_tracer_amd", "csrc", "bt_api.cpp")).read()
    assert "getenv" not in src


def test_default_sample_base_never_replays_indices(bendy):
    """Tracer.render's default sample_base is the next UNUSED sample index: after 3 calls with Subsample::None a call with
    Subpixel(2) must start at index ceil(3 / 4) = 1 of its own numbering, not at 0."""
    buf = bendy.Buffer.new(4, 4, device="cpu")
    buf.inc_samples(3)
    nn = bendy.Subsample(2).subpixel_count()
    assert (buf.samples + nn - 1) // nn == 1
