"""ctypes binding + independent Python scene loader for the CPU oracle.

TEST INFRASTRUCTURE ONLY (see oracle/bt_oracle.h).  Only tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.

The scene loader here (gzip + json from the standard library) is deliberately
independent of the product's C++ loader (bendy_tracer_amd/csrc/bt_scene.cpp) so
that tests can cross-check the two.  File format: serde externally-tagged enums,
SURVEY.md 8(b-2); reference types scene/mod.rs:16-20,84-90, object/mod.rs:23-41,
247-256, data/mod.rs:12-51, material.rs:22-44, volume.rs:75-82.
"""
from __future__ import annotations

import ctypes as C
import gzip
import json
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libbt_oracle.so")

EMPTY, CAMERA, SPHERE, RECT, CUBOID = range(5)
FLAT, DIFFUSE, METALLIC, GLASS, EMISSIVE, VOLUME = range(6)
OUT_FULL, OUT_ALBEDO, OUT_NORMAL, OUT_DEPTH = range(4)
FACE_NAMES = ["Front", "Back", "Volume", "VolumeFront", "VolumeBack"]


class V3(C.Structure):
    _fields_ = [("x", C.c_float), ("y", C.c_float), ("z", C.c_float)]


class Affine(C.Structure):
    _fields_ = [("cx", V3), ("cy", V3), ("cz", V3), ("t", V3)]


class Rect(C.Structure):
    _fields_ = [("material", C.c_int32), ("half_width", C.c_float), ("half_height", C.c_float),
                ("x", V3), ("y", V3), ("z", V3)]


class Object(C.Structure):
    _fields_ = [("object_ref", C.c_uint64), ("kind", C.c_int32), ("flags", C.c_uint32), ("world", Affine),
                ("sensor_size", C.c_float), ("focal_length", C.c_float), ("aspect_ratio", C.c_float),
                ("fstop", C.c_float), ("focus", C.c_float), ("has_focus", C.c_int32),
                ("material", C.c_int32), ("volume", C.c_int32), ("radius", C.c_float),
                ("rect", Rect), ("face_offset", V3 * 6), ("faces", Rect * 6)]


class Data(C.Structure):
    _fields_ = [("data_ref", C.c_uint64), ("kind", C.c_int32), ("albedo", C.c_float * 3),
                ("roughness", C.c_float), ("ior", C.c_float), ("intensity", C.c_float),
                ("width", C.c_int32), ("height", C.c_int32), ("depth", C.c_int32),
                ("size", C.c_float * 3), ("buffer_offset", C.c_int64)]


class SceneC(C.Structure):
    _fields_ = [("n_objects", C.c_int32), ("n_data", C.c_int32), ("objects", C.POINTER(Object)),
                ("data", C.POINTER(Data)), ("density", C.POINTER(C.c_float)), ("root_material", C.c_int32)]


class Config(C.Structure):
    """tracer/mod.rs Config (:16-45) merged with RenderConfig (:117-135)."""
    _fields_ = [("max_bounces", C.c_int32), ("max_volume_bounces", C.c_int32), ("clip_min", C.c_float),
                ("clip_max", C.c_float), ("volume_step", C.c_float), ("chunks_x", C.c_int32),
                ("chunks_y", C.c_int32), ("output", C.c_int32), ("samples", C.c_int32),
                ("subsample_n", C.c_int32), ("sample_base", C.c_uint32), ("recursive", C.c_int32),
                # lens extension (NOT in the reference; default off)
                ("lens_on", C.c_int32), ("lens_centre", C.c_float * 3), ("lens_rs", C.c_float),
                ("lens_step", C.c_float), ("lens_radius", C.c_float), ("lens_max_steps", C.c_int32)]


def default_config(samples=1, subsample_n=0, output=OUT_FULL, recursive=1, chunks=(8, 4), sample_base=0,
                   max_bounces=8, max_volume_bounces=None, volume_step=0.1, lens=None):
    """Config::DEFAULT (mod.rs:29-38) with main.rs's 8x4 chunks (main.rs:225-230).
    Q1 (mod.rs:224): a RenderConfig.max_bounces override also overrides max_volume_bounces;
    without an override max_volume_bounces stays 32.
    lens = dict(centre=(x, y, z), rs=, step=, radius=, max_steps=) switches the lens extension on."""
    cfg = Config(max_bounces=max_bounces, max_volume_bounces=32 if max_volume_bounces is None else max_volume_bounces,
                 clip_min=0.01, clip_max=1000.0, volume_step=volume_step, chunks_x=chunks[0], chunks_y=chunks[1],
                 output=output, samples=samples, subsample_n=subsample_n, sample_base=sample_base,
                 recursive=recursive)
    if lens is not None:
        cfg.lens_on = 1
        cfg.lens_centre[0], cfg.lens_centre[1], cfg.lens_centre[2] = lens["centre"]
        cfg.lens_rs, cfg.lens_step, cfg.lens_radius = lens["rs"], lens["step"], lens["radius"]
        cfg.lens_max_steps = lens.get("max_steps", 4096)
    return cfg


def lens_trace_free(cfg, origin, direction):
    """Lens extension in empty space: (status, end position, end direction); status 0 escaped, 1 captured, 2 budget."""
    pos, d = (C.c_float * 3)(), (C.c_float * 3)()
    st = lib().bto_lens_trace_free(C.byref(cfg), _f3(origin), _f3(direction), pos, d)
    return st, np.array(pos[:], np.float32), np.array(d[:], np.float32)


def build(force=False):
    if force or not os.path.exists(_LIB_PATH) or (
            os.path.getmtime(_LIB_PATH) < os.path.getmtime(os.path.join(_HERE, "bt_oracle.c"))):
        subprocess.check_call(["make", "-C", _HERE, "-B" if force else "-s"], stdout=subprocess.DEVNULL)
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_LIB_PATH)
        fp = C.POINTER(C.c_float)
        L.bto_render.restype = C.c_int
        L.bto_render.argtypes = [C.POINTER(SceneC), C.c_int32, C.POINTER(Config), fp, C.c_uint32, C.c_uint32,
                                 C.c_uint64, C.c_int32, C.POINTER(C.c_uint64)]
        L.bto_trace_one.restype = C.c_int
        L.bto_trace_one.argtypes = [C.POINTER(SceneC), C.c_int32, C.POINTER(Config), C.c_uint32, C.c_uint32,
                                    C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint64, fp]
        L.bto_philox4x32_10.argtypes = [C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
        L.bto_sincos.argtypes = [C.c_float, fp, fp]
        L.bto_uniform_scale.restype = C.c_float
        L.bto_uniform_scale.argtypes = [C.c_float, C.c_float, C.c_int]
        L.bto_ray_with_frustum.argtypes = [C.c_float, C.c_float, C.c_float, C.c_float, fp]
        L.bto_object_hit.restype = C.c_int
        L.bto_object_hit.argtypes = [C.POINTER(SceneC), C.c_int32, fp, fp, C.c_float, C.c_float, C.c_int, fp, fp, fp]
        L.bto_object_pdf.restype = C.c_float
        L.bto_object_pdf.argtypes = [C.POINTER(SceneC), C.c_int32, fp, fp, C.c_float, C.c_float]
        L.bto_reflect.argtypes = [fp, fp, fp]
        L.bto_refract.argtypes = [fp, fp, C.c_float, fp]
        L.bto_fresnel.restype = C.c_float
        L.bto_fresnel.argtypes = [fp, fp, C.c_float]
        L.bto_density_sample.restype = C.c_float
        L.bto_density_sample.argtypes = [C.POINTER(SceneC), C.c_int32, fp]
        L.bto_orthonormal_pair.argtypes = [fp, fp, fp]
        L.bto_affine_inverse.argtypes = [C.POINTER(Affine), C.POINTER(Affine)]
        L.bto_preview.argtypes = [fp, C.c_uint32, C.c_uint32, C.c_int32, C.POINTER(C.c_uint8)]
        L.bto_chunk_bounds.argtypes = [C.c_uint32, C.c_uint32, C.c_int32, C.c_int32, C.POINTER(C.c_int32),
                                       C.POINTER(C.c_uint32)]
        L.bto_lens_trace_free.restype = C.c_int
        L.bto_lens_trace_free.argtypes = [C.POINTER(Config), fp, fp, fp, fp]
        _lib = L
    return _lib


# ----------------------------------------------------------------------------- scene loading
def load_scene_json(path):
    """main.rs:93-102: gzip iff the extension is .gz."""
    opener = gzip.open if str(path).endswith(".gz") else open
    with opener(path, "rt") as f:
        return json.load(f)


def _v3(a):
    return V3(float(a[0]), float(a[1]), float(a[2]))


def _affine(a):
    return Affine(_v3(a[0:3]), _v3(a[3:6]), _v3(a[6:9]), _v3(a[9:12]))


class Scene:
    """Flat, ctypes-backed view of a scene.json document for the oracle.

    Objects are ordered by ascending ObjectRef (the reference iterates a hashbrown
    map whose order is unspecified; DESIGN.md Q11 fixes ascending order)."""

    def __init__(self, doc):
        self.doc = doc
        data_keys = sorted(int(k) for k in doc["data"]["collection"])
        self.data_index = {k: i for i, k in enumerate(data_keys)}
        obj_keys = sorted(int(k) for k in doc["objects"]["collection"])
        self.object_index = {k: i for i, k in enumerate(obj_keys)}
        self.object_keys = obj_keys
        self.tags = {}

        density = []
        self._data = (Data * max(1, len(data_keys)))()
        for i, k in enumerate(data_keys):
            inner = doc["data"]["collection"][str(k)]["inner"]
            d = self._data[i]
            d.data_ref = k
            if "Material" in inner:
                (name, m), = inner["Material"].items()
                d.kind = {"Flat": FLAT, "Diffuse": DIFFUSE, "Metallic": METALLIC, "Glass": GLASS,
                          "Emissive": EMISSIVE}[name]
                a = m["albedo"]
                d.albedo[0], d.albedo[1], d.albedo[2] = a["r"], a["g"], a["b"]
                d.roughness = m.get("roughness", 0.0)
                d.ior = m.get("ior", 1.0)
                d.intensity = m.get("intensity", 0.0)
            else:
                dm = inner["Volume"]["DensityMap"]
                d.kind = VOLUME
                d.width, d.height, d.depth = dm["width"], dm["height"], dm["depth"]
                d.size[0], d.size[1], d.size[2] = dm["size"]
                d.buffer_offset = len(density)
                density.extend(dm["buffer"])
        self._density = np.asarray(density if density else [0.0], dtype=np.float32)

        def mat(ref):
            return self.data_index[int(ref)]

        def rect(r):
            return Rect(mat(r["material"]), r["half_width"], r["half_height"], _v3(r["x"]), _v3(r["y"]), _v3(r["z"]))

        self._objects = (Object * max(1, len(obj_keys)))()
        for i, k in enumerate(obj_keys):
            src = doc["objects"]["collection"][str(k)]
            o = self._objects[i]
            o.object_ref = k
            o.flags = src["flags"]["bits"]
            o.world = _affine(src["transform"]["transform_world"])
            o.material = -1
            o.volume = -1
            if src.get("tag") is not None:
                self.tags.setdefault(src["tag"], i)
            inner = src["inner"]
            if inner == "Empty":
                o.kind = EMPTY
                continue
            (name, body), = inner.items()
            if name == "Camera":
                o.kind = CAMERA
                o.sensor_size, o.focal_length = body["sensor_size"], body["focal_length"]
                o.aspect_ratio, o.fstop = body["aspect_ratio"], body["fstop"]
                o.has_focus = body["focus"] is not None
                o.focus = body["focus"] if body["focus"] is not None else 0.0
            elif name == "Sphere":
                o.kind = SPHERE
                o.material = mat(body["material"])
                o.volume = mat(body["volume"]) if body["volume"] is not None else -1
                o.radius = body["radius"]
            elif name == "Rect":
                o.kind = RECT
                o.rect = rect(body)
            elif name == "Cuboid":
                o.kind = CUBOID
                for f, (offset, r) in enumerate(body["faces"]):
                    o.face_offset[f] = _v3(offset)
                    o.faces[f] = rect(r)
            else:
                raise ValueError(f"unknown object kind {name}")

        self.c = SceneC(len(obj_keys), len(data_keys), self._objects, self._data,
                        self._density.ctypes.data_as(C.POINTER(C.c_float)), self.data_index[int(doc["root_material"])])

    @classmethod
    def load(cls, path):
        return cls(load_scene_json(path))

    def find_by_tag(self, tag):
        """Scene::find_by_tag, scene/mod.rs:124-129 -> object index."""
        return self.tags.get(tag)

    def set_camera_aspect(self, cam_index, aspect):
        """main.rs:218-223 (Q12)."""
        self._objects[cam_index].aspect_ratio = aspect

    def n_lights(self):
        return sum(1 for i in range(self.c.n_objects) if self._objects[i].flags & 1)


def render(scene: Scene, camera: int, cfg: Config, width: int, height: int, seed: int, nthreads: int = 1,
           rgba: np.ndarray | None = None):
    """Tracer::render (mod.rs:179-202) on a Buffer::new-style accumulator (buffer.rs:41-50).
    Returns (rgba, status, segments)."""
    if rgba is None:
        rgba = np.zeros((height, width, 4), dtype=np.float32)
        rgba[..., 3] = 1.0
    seg = C.c_uint64(0)
    rc = lib().bto_render(C.byref(scene.c), camera, C.byref(cfg), rgba.ctypes.data_as(C.POINTER(C.c_float)), width,
                          height, seed, nthreads, C.byref(seg))
    if rc < 0:
        raise RuntimeError(f"bto_render failed: {rc}")
    return rgba, rc, seg.value


def trace_one(scene: Scene, camera: int, cfg: Config, width: int, height: int, px: int, py: int, sample_index: int,
              seed: int):
    out = (C.c_float * 10)()
    rc = lib().bto_trace_one(C.byref(scene.c), camera, C.byref(cfg), width, height, px, py, sample_index, seed, out)
    if rc < 0:
        raise RuntimeError(f"bto_trace_one failed: {rc}")
    a = np.array(out[:], dtype=np.float32)
    return {"color": a[0:3], "albedo": a[3:6], "normal": a[6:9], "depth": a[9]}


def _f3(a):
    return (C.c_float * 3)(*[float(v) for v in a])


def object_hit(scene: Scene, index: int, origin, direction, clip=(0.01, 1000.0), volumetric=False):
    t = C.c_float()
    pos, nrm = (C.c_float * 3)(), (C.c_float * 3)()
    face = lib().bto_object_hit(C.byref(scene.c), index, _f3(origin), _f3(direction), clip[0], clip[1],
                                int(volumetric), C.byref(t), pos, nrm)
    if face < 0:
        return None
    return {"face": FACE_NAMES[face], "t": t.value, "position": np.array(pos[:], np.float32),
            "normal": np.array(nrm[:], np.float32)}


def object_pdf(scene: Scene, index: int, origin, direction, clip=(0.01, 1000.0)):
    p = lib().bto_object_pdf(C.byref(scene.c), index, _f3(origin), _f3(direction), clip[0], clip[1])
    return None if p < 0 else p


def philox(ctr, key):
    c = (C.c_uint32 * 4)(*ctr)
    k = (C.c_uint32 * 2)(*key)
    o = (C.c_uint32 * 4)()
    lib().bto_philox4x32_10(c, k, o)
    return list(o)


def sincos(x):
    s, c = C.c_float(), C.c_float()
    lib().bto_sincos(x, C.byref(s), C.byref(c))
    return s.value, c.value


def preview(rgba: np.ndarray, samples: int, color_space: int = 3):
    flat = np.ascontiguousarray(rgba, dtype=np.float32).reshape(-1, 4)
    out = np.zeros((flat.shape[0], 4), dtype=np.uint8)
    lib().bto_preview(flat.ctypes.data_as(C.POINTER(C.c_float)), flat.shape[0], samples, color_space,
                      out.ctypes.data_as(C.POINTER(C.c_uint8)))
    return out.reshape(rgba.shape[:-1] + (4,))


def chunk_bounds(width, height, cx, cy):
    n = C.c_int32()
    lib().bto_chunk_bounds(width, height, cx, cy, C.byref(n), None)
    b = (C.c_uint32 * (4 * n.value))()
    lib().bto_chunk_bounds(width, height, cx, cy, C.byref(n), b)
    return np.array(b[:], dtype=np.uint32).reshape(-1, 4)
