"""the callers' side -- scene save (pretty JSON / gzip), the built-in default scene,
PNG screenshots and the headless CLI that mirrors src/main.rs."""
import gzip
import json
import os
import struct
import subprocess
import zlib

import numpy as np
import pytest

from conftest import ROOT, scene_path

CLI = os.path.join(ROOT, "bendy_tracer_amd", "bendy-tracer-hip")
ALL_SCENES = ["scene", "cornell", "cornell2", "volume", "cloud"]


def read_png(path):
    raw = open(path, "rb").read()
    assert raw[:8] == b"\x89PNG\r\n\x1a\n"
    pos, idat, w = 8, b"", None
    while pos < len(raw):
        n, typ = struct.unpack(">I4s", raw[pos:pos + 8])
        body = raw[pos + 8:pos + 8 + n]
        assert struct.unpack(">I", raw[pos + 8 + n:pos + 12 + n])[0] == zlib.crc32(typ + body) & 0xFFFFFFFF
        if typ == b"IHDR":
            w, h, depth, ctype = struct.unpack(">IIBB", body[:10])
            assert (depth, ctype) == (8, 6)
        elif typ == b"IDAT":
            idat += body
        pos += 12 + n
    rows = np.frombuffer(zlib.decompress(idat), np.uint8).reshape(h, 1 + w * 4)
    assert (rows[:, 0] == 0).all()
    return rows[:, 1:].reshape(h, w, 4)


@pytest.mark.parametrize("name", ALL_SCENES)
def test_scene_json_round_trip(bendy, name, tmp_path):
    """serde_json::to_writer_pretty (main.rs:299-313): same document, two-space indent; reload is bit-identical."""
    raw = gzip.open(scene_path(name)).read()
    sc = bendy.Scene.load(scene_path(name))
    txt = sc.to_json()
    assert json.loads(txt) == json.loads(raw)
    assert txt.startswith('{\n  "roots": [],\n  "root_material": ')
    assert len(txt.splitlines()) == len(json.dumps(json.loads(raw), indent=2).splitlines())
    for ext in (".json", ".json.gz"):
        p = tmp_path / ("saved" + ext)
        sc.save(p)
        data = open(p, "rb").read()
        assert (data[:2] == b"\x1f\x8b") == ext.endswith(".gz")      # main.rs:305: gzip iff the extension is gz
        back = bendy.Scene.load(p)
        assert np.array_equal(back.export_prims().view(np.uint32), sc.export_prims().view(np.uint32))
        assert back.to_json() == txt


def test_saved_scene_carries_the_camera_aspect(bendy, tmp_path):
    sc = bendy.Scene.load(scene_path("scene"))
    cam = sc.find_by_tag("camera")
    sc.set_camera_aspect(cam, 768 / 512)                 # main.rs:218-223 mutates the scene before any save
    doc = json.loads(sc.to_json())
    assert doc["objects"]["collection"][str(cam)]["inner"]["Camera"]["aspect_ratio"] == 1.5
    orig = json.loads(gzip.open(scene_path("scene")).read())
    doc["objects"]["collection"][str(cam)]["inner"]["Camera"]["aspect_ratio"] = 1.7777778
    assert doc == orig                                   # nothing else changed


def test_default_scene_is_the_cornell_box_of_main_rs(bendy):
    """main.rs:107-214; the bundled cornell2.json.gz is that scene saved by the reference."""
    d, c = bendy.Scene.default(), bendy.Scene.load(scene_path("cornell2"))
    pd, pc = d.export_prims(), c.export_prims()
    assert pd.shape == pc.shape == (18, 36)
    assert np.array_equal(pd[:, :4].view(np.int32), pc[:, :4].view(np.int32))
    assert np.nanmax(np.abs(pd[:, 4:] - pc[:, 4:])) < 1e-6     # tall box rotation: libm vs glam quaternion, 1 ulp
    dd, cc = json.loads(d.to_json()), json.loads(gzip.open(scene_path("cornell2")).read())
    assert dd["data"] == cc["data"] and dd["root_material"] == cc["root_material"]
    assert d.find_by_tag("camera") == 0
    # cornell2.json.gz was saved from a square window, so its camera carries aspect 1.0 (main.rs:218-223);
    # a fresh default scene has Camera::DEFAULT's 1.5 (camera.rs:12-20)
    assert dd["objects"]["collection"]["0"]["inner"]["Camera"]["aspect_ratio"] == 1.5
    dd["objects"]["collection"]["0"]["inner"]["Camera"]["aspect_ratio"] = 1.0
    for k in "01234568":                                    # everything but the rotated tall box is identical
        assert dd["objects"]["collection"][k] == cc["objects"]["collection"][k], k
    assert dd["objects"]["collection"]["7"]["inner"] == cc["objects"]["collection"]["7"]["inner"]


def test_write_png(bendy, tmp_path):
    rng = np.random.default_rng(0)
    img = rng.integers(0, 256, (37, 53, 4), dtype=np.uint8)
    p = tmp_path / "x.png"
    bendy.write_png(p, img)
    assert np.array_equal(read_png(p), img)


def test_cli_argument_errors():
    r = subprocess.run([CLI, "--width", "8"], capture_output=True, text=True)
    assert r.returncode != 0 and "--output" in r.stderr          # clap: required argument (main.rs:57-58)
    r = subprocess.run([CLI, "--output", "depth"], capture_output=True, text=True)
    assert r.returncode != 0 and "possible values: full, albedo, normal" in r.stderr
    r = subprocess.run([CLI, "--help"], capture_output=True, text=True)
    assert r.returncode == 0 and "--screenshot" in r.stderr


def test_cli_fails_loudly_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    r = subprocess.run([CLI, "--output", "full", "--width", "8", "--height", "8"], capture_output=True, text=True)
    assert r.returncode != 0 and "error" in r.stderr


@pytest.mark.gpu
@pytest.mark.parametrize("output,scene", [("full", "scene"), ("normal", "cornell2"), ("albedo", "volume"), ("full", None)])
def test_cli_renders_like_the_library(bendy, tmp_path, output, scene):
    """The CLI's progressive loop (1 x subsample^2 rays per call, main.rs:245-254) and screenshot equal
    the same loop driven through the Python mirror of the API, pixel for pixel."""
    import torch
    w, h, samples, sub = 96, 64, 8, 2
    shot = tmp_path / "shots" / "nested" / "out.png"
    saved = tmp_path / "saved.json.gz"
    cmd = [CLI, "--output", output, "--width", str(w), "--height", str(h), "--samples", str(samples), "--subsample", str(sub),
           "--screenshot", str(shot), "--scene", scene_path(scene) if scene else str(tmp_path / "absent.json"),
           "--save-scene", str(saved), "--seed", "77"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    assert f"samples: {samples}/{samples}" in r.stderr and "saved screenshot to" in r.stderr
    assert ("loaded scene from" in r.stderr) == (scene is not None)

    sc = bendy.Scene.load(scene_path(scene)) if scene else bendy.Scene.default()
    cam = sc.find_by_tag("camera")
    sc.set_camera_aspect(cam, w / h)
    out = {"full": bendy.Output.Full, "albedo": bendy.Output.Albedo, "normal": bendy.Output.Normal}[output]
    cs = bendy.ColorSpace.Normal if output == "normal" else bendy.ColorSpace.SRgb       # main.rs:40-46
    buf = bendy.Buffer.new(w, h, cs)
    tr = bendy.Tracer.with_config(bendy.Config(chunks_x=8, chunks_y=4, output=out))
    while buf.samples < samples:
        tr.render(sc, cam, bendy.RenderConfig.with_samples_subsample(1, bendy.Subsample(sub)), buf, seed=77)
    torch.cuda.synchronize()
    assert np.array_equal(read_png(shot), buf.preview())
    back = bendy.Scene.load(saved)                                   # Ctrl+K output reloads to the same scene
    assert np.array_equal(back.export_prims().view(np.uint32), sc.export_prims().view(np.uint32))


def _build_hpp_smoke(tmp_path):
    exe = tmp_path / "hpp_smoke"
    libdir = os.path.join(ROOT, "bendy_tracer_amd")
    subprocess.check_call(["g++", "-std=c++20", "-O1", "-Wall", "-I" + os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "cpp", "hpp_smoke.cpp"), "-L" + libdir, "-lbendy_hip",
                           "-Wl,-rpath," + libdir, "-o", str(exe)])
    return exe


def test_cpp_header_mirror_without_gpu(tmp_path):
    """include/bendy_tracer.hpp (Scene / Tracer / Buffer / RenderConfig ...) compiles with plain g++ against
    the C ABI; defaults, loader errors and Status::Done work without a device, render fails loudly."""
    import torch
    exe = _build_hpp_smoke(tmp_path)
    r = subprocess.run([str(exe), scene_path("scene"), str(tmp_path / "o.bin")], capture_output=True, text=True)
    if torch.cuda.is_available():
        assert r.returncode == 0, r.stdout + r.stderr
    else:
        assert r.returncode == 42 and "render error -8" in r.stdout      # BT_ERR_DEVICE, no CPU fallback


@pytest.mark.gpu
def test_cpp_header_mirror_renders_like_python(bendy, tmp_path):
    import torch
    exe = _build_hpp_smoke(tmp_path)
    out = tmp_path / "o.bin"
    r = subprocess.run([str(exe), scene_path("scene"), str(out)], capture_output=True, text=True)
    assert r.returncode == 0 and "ok samples=8" in r.stdout, r.stdout + r.stderr
    w, h = 48, 27
    raw = np.fromfile(out, dtype=np.uint8)
    sums = raw[:w * h * 16].view(np.float32).reshape(h, w, 4)
    rgba8 = raw[w * h * 16:].reshape(h, w, 4)
    sc = bendy.Scene.load(scene_path("scene")); cam = sc.find_by_tag("camera"); sc.set_camera_aspect(cam, w / h)
    buf = bendy.Buffer.new(w, h, bendy.ColorSpace.SRgb)
    tr = bendy.Tracer.with_config(bendy.Config(chunks_x=8, chunks_y=4))
    while buf.samples < 8:
        tr.render(sc, cam, bendy.RenderConfig.with_samples_subsample(1, bendy.Subsample(2)), buf, seed=99)
    torch.cuda.synchronize()
    assert np.array_equal(sums, buf.numpy()) and np.array_equal(rgba8, buf.preview())


@pytest.mark.gpu
def test_cli_screenshot_without_extension_uses_render_png(tmp_path):
    d = tmp_path / "shots"
    r = subprocess.run([CLI, "--output", "full", "--width", "32", "--height", "16", "--samples", "1", "--subsample", "1",
                        "--screenshot", str(d / "noext"), "--scene", scene_path("scene"), "--quiet"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert (d / "render.png").exists()                               # main.rs:277-281 DEFAULT_SCREENSHOT
