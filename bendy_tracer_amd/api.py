"""Host-side mirror of the reference's library API over the C ABI of libbendy_hip.so.

The reference is a Rust crate; `src/main.rs` drives it through `Scene`, `Tracer`,
`Config`, `RenderConfig`, `Subsample`, `Output`, `Status`, `Buffer`, `ColorSpace`
(tracer/mod.rs:16-203, tracer/buffer.rs:11-179, scene/mod.rs:84-146).  Rust is not
available in this image, so this module re-exposes the same names, argument meaning and
error behaviour in Python (panics become exceptions), calling the HIP implementation
through `include/bendy_hip.h`.  There is NO CPU fallback: importing this module fails if
the shared library has not been built, and rendering fails without a gfx950 device.
"""
from __future__ import annotations

import ctypes as C
import enum
import os
from dataclasses import dataclass
from typing import Optional

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libbendy_hip.so")
BT_TILE = 16


class BendyError(RuntimeError):
    """A reference panic / serde error, surfaced as a negative bt_status."""

    def __init__(self, code, message):
        super().__init__(f"[bt_status {code}] {message}")
        self.code = code


class Output(enum.IntEnum):  # tracer/mod.rs:108-115
    Full = 0
    Albedo = 1
    Normal = 2
    Depth = 3


class ColorSpace(enum.IntEnum):  # tracer/buffer.rs:11-17
    NONE = 0
    Normal = 1
    Linear = 2
    SRgb = 3


class Status(enum.IntEnum):  # tracer/mod.rs:159-163
    Done = 0
    InProgress = 1


@dataclass(frozen=True)
class Subsample:  # tracer/mod.rs:47-106
    n: int = 0  # 0 = Subsample::None, n = Subsample::Subpixel(n)

    @staticmethod
    def none():
        return Subsample(0)

    @staticmethod
    def subpixel(n):
        return Subsample(int(n))

    def subpixel_size(self):  # :55-60
        return 1.0 if self.n == 0 else float(np.float32(1.0) / np.float32(self.n))

    def subpixel_count(self):  # :62-67
        return 1 if self.n == 0 else self.n * self.n

    def __iter__(self):  # :87-106: (i/n, j/n), i fastest
        if self.n == 0:
            yield (0.0, 0.0)
            return
        w = np.float32(1.0) / np.float32(self.n)
        for c in range(self.n * self.n):
            yield (float(np.float32(c % self.n) * w), float(np.float32(c // self.n) * w))


@dataclass
class Config:  # tracer/mod.rs:16-45
    max_bounces: int = 8
    max_volume_bounces: int = 32
    clip_min: float = 0.01
    clip_max: float = 1000.0
    volume_step: float = 0.1
    chunks_x: int = 4
    chunks_y: int = 2
    output: Output = Output.Full


@dataclass
class RenderConfig:  # tracer/mod.rs:117-157
    subsample: Subsample = Subsample(0)
    samples: int = 64
    output: Optional[Output] = None
    max_bounces: Optional[int] = None
    max_volume_bounces: Optional[int] = None
    volume_step: Optional[float] = None

    @staticmethod
    def with_samples(samples):  # :137-142
        return RenderConfig(samples=samples)

    @staticmethod
    def with_samples_subsample(samples, subsample):  # :144-150
        return RenderConfig(samples=samples, subsample=subsample)


class _CConfig(C.Structure):
    _fields_ = [("max_bounces", C.c_uint32), ("max_volume_bounces", C.c_uint32), ("clip_min", C.c_float),
                ("clip_max", C.c_float), ("volume_step", C.c_float), ("chunks_x", C.c_uint32),
                ("chunks_y", C.c_uint32), ("output", C.c_int32)]


class _CRenderConfig(C.Structure):
    _fields_ = [("subsample_n", C.c_uint32), ("samples", C.c_uint32), ("has_output", C.c_int32),
                ("output", C.c_int32), ("has_max_bounces", C.c_int32), ("max_bounces", C.c_uint32),
                ("has_max_volume_bounces", C.c_int32), ("max_volume_bounces", C.c_uint32),
                ("has_volume_step", C.c_int32), ("volume_step", C.c_float), ("sample_base", C.c_uint32)]


class Stats(C.Structure):
    _fields_ = [("samples", C.c_uint64), ("segments", C.c_uint64), ("pixels", C.c_uint64), ("kernel_ms", C.c_float),
                ("lens_steps", C.c_uint64), ("slices", C.c_uint32), ("launches", C.c_uint32),
                ("scratch_bytes", C.c_uint64), ("parked_bytes", C.c_uint64), ("workgroups", C.c_uint32), ("packed", C.c_uint32)]


class _CTuning(C.Structure):  # include/bendy_hip.h `bt_tuning`
    _fields_ = [("slices", C.c_uint32), ("phase_vote", C.c_int32), ("scratch_cap_bytes", C.c_uint64), ("packed", C.c_int32),
                ("reserved", C.c_int32)]


class _CLens(C.Structure):
    _fields_ = [("centre", C.c_float * 3), ("rs", C.c_float), ("step", C.c_float), ("radius", C.c_float),
                ("max_steps", C.c_uint32)]


EXPORTS = [
    "bt_config_default", "bt_render_config_default", "bt_last_error", "bt_last_error_code", "bt_version", "bt_scene_load",
    "bt_scene_from_json", "bt_scene_free", "bt_scene_find_by_tag", "bt_scene_set_camera_aspect", "bt_scene_set_lens",
    "bt_scene_object_count", "bt_scene_data_count", "bt_scene_export_prims", "bt_render", "bt_render_device",
    "bt_shard_floats", "bt_render_shard_device", "bt_unshard_device", "bt_preview_device", "bt_preview",
    "bt_comm_unique_id", "bt_comm_init", "bt_comm_free", "bt_comm_rank", "bt_comm_world", "bt_allgather_shards_device",
    "bt_exchange_frame_device", "bt_scene_last_stats", "bt_tuning_default", "bt_scene_set_tuning", "bt_scene_get_tuning", "bt_scene_default", "bt_scene_to_json", "bt_scene_save", "bt_write_png",
    "bt_scene_trim",
]


def _load():
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: the HIP extension has not been built (run `python -c 'import __graft_entry__ "
            "as g; g.build()'` or `make -C bendy_tracer_amd/csrc`).  There is no CPU fallback.")
    # PyTorch bundles its own HIP / HSA runtime.  Two copies of the runtime in one process cannot both own
    # the GPU (the second one reports "no ROCm-capable device"), so when torch is installed it is imported
    # FIRST and libbendy_hip.so then binds to the runtime torch has already loaded.  Standalone C/C++ users
    # (the CLI) use /opt/rocm's runtime.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    L = C.CDLL(LIB_PATH)
    vp, fp = C.c_void_p, C.POINTER(C.c_float)
    L.bt_last_error.restype = C.c_char_p
    L.bt_version.restype = C.c_char_p
    L.bt_config_default.argtypes = [C.POINTER(_CConfig)]
    L.bt_render_config_default.argtypes = [C.POINTER(_CRenderConfig)]
    L.bt_scene_load.restype = vp
    L.bt_scene_load.argtypes = [C.c_char_p]
    L.bt_scene_from_json.restype = vp
    L.bt_scene_from_json.argtypes = [C.c_char_p, C.c_size_t]
    L.bt_scene_free.argtypes = [vp]
    L.bt_scene_default.restype = vp
    L.bt_scene_to_json.argtypes = [vp, C.c_char_p, C.c_size_t]
    L.bt_scene_save.argtypes = [vp, C.c_char_p]
    L.bt_write_png.argtypes = [C.c_char_p, C.POINTER(C.c_uint8), C.c_uint32, C.c_uint32]
    L.bt_scene_find_by_tag.argtypes = [vp, C.c_char_p, C.POINTER(C.c_uint64)]
    L.bt_scene_set_camera_aspect.argtypes = [vp, C.c_uint64, C.c_float]
    L.bt_scene_set_lens.argtypes = [vp, C.POINTER(_CLens)]
    L.bt_scene_object_count.argtypes = [vp]
    L.bt_scene_data_count.argtypes = [vp]
    L.bt_scene_export_prims.argtypes = [vp, fp, C.c_int]
    L.bt_render.argtypes = [vp, C.c_uint64, C.POINTER(_CConfig), C.POINTER(_CRenderConfig), fp, C.c_uint32,
                            C.c_uint32, C.c_uint64]
    L.bt_render_device.argtypes = [vp, C.c_uint64, C.POINTER(_CConfig), C.POINTER(_CRenderConfig), vp, C.c_uint32,
                                   C.c_uint32, C.c_uint64, vp]
    L.bt_shard_floats.restype = C.c_size_t
    L.bt_shard_floats.argtypes = [C.c_uint32, C.c_uint32, C.c_uint32]
    L.bt_render_shard_device.argtypes = [vp, C.c_uint64, C.POINTER(_CConfig), C.POINTER(_CRenderConfig), vp,
                                         C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint64, vp]
    L.bt_unshard_device.argtypes = [vp, vp, C.c_uint32, C.c_uint32, C.c_uint32, vp]
    L.bt_preview_device.argtypes = [vp, vp, C.c_uint32, C.c_uint32, C.c_uint32, C.c_int32, vp]
    L.bt_preview.argtypes = [fp, C.POINTER(C.c_uint8), C.c_uint32, C.c_uint32, C.c_uint32, C.c_int32]
    L.bt_scene_last_stats.argtypes = [vp, C.POINTER(Stats)]
    L.bt_comm_unique_id.argtypes = [C.c_void_p, C.c_size_t]
    L.bt_comm_init.restype = vp
    L.bt_comm_init.argtypes = [C.c_uint32, C.c_uint32, C.c_void_p, C.c_size_t]
    L.bt_comm_free.argtypes = [vp]
    L.bt_comm_rank.argtypes = [vp]
    L.bt_comm_world.argtypes = [vp]
    L.bt_allgather_shards_device.argtypes = [vp, vp, vp, C.c_uint32, C.c_uint32, vp]
    L.bt_exchange_frame_device.argtypes = [vp, vp, vp, vp, C.c_uint32, C.c_uint32, vp]
    L.bt_tuning_default.argtypes = [C.POINTER(_CTuning)]
    L.bt_scene_set_tuning.argtypes = [vp, C.POINTER(_CTuning)]
    L.bt_scene_get_tuning.argtypes = [vp, C.POINTER(_CTuning)]
    L.bt_scene_trim.argtypes = [vp]
    return L


lib = _load()


def _check(rc):
    if rc < 0:
        raise BendyError(rc, lib.bt_last_error().decode("utf-8", "replace"))
    return rc


def _c_configs(config: Config, render: RenderConfig, sample_base: int):
    c = _CConfig(config.max_bounces, config.max_volume_bounces, config.clip_min, config.clip_max, config.volume_step,
                 config.chunks_x, config.chunks_y, int(config.output))
    r = _CRenderConfig()
    r.subsample_n = render.subsample.n
    r.samples = render.samples
    r.has_output = render.output is not None
    r.output = int(render.output) if render.output is not None else 0
    r.has_max_bounces = render.max_bounces is not None
    r.max_bounces = render.max_bounces or 0
    r.has_max_volume_bounces = render.max_volume_bounces is not None
    r.max_volume_bounces = render.max_volume_bounces or 0
    r.has_volume_step = render.volume_step is not None
    r.volume_step = render.volume_step or 0.0
    r.sample_base = sample_base
    return c, r


class Scene:
    """`Scene` (scene/mod.rs:84-146) as loaded by main.rs:93-102."""

    def __init__(self, handle):
        if not handle:
            raise BendyError(lib.bt_last_error_code(), lib.bt_last_error().decode("utf-8", "replace"))
        self._h = C.c_void_p(handle)

    @classmethod
    def load(cls, path):
        return cls(lib.bt_scene_load(os.fspath(path).encode()))

    @classmethod
    def from_json(cls, text):
        data = text.encode() if isinstance(text, str) else bytes(text)
        return cls(lib.bt_scene_from_json(data, len(data)))

    @classmethod
    def default(cls):
        """The built-in Cornell scene of main.rs:107-214."""
        return cls(lib.bt_scene_default())

    def to_json(self) -> str:
        """serde_json::to_string_pretty(&scene)."""
        n = _check(lib.bt_scene_to_json(self._h, None, 0))
        buf = C.create_string_buffer(n + 1)
        _check(lib.bt_scene_to_json(self._h, buf, n + 1))
        return buf.value.decode()

    def save(self, path):
        """Ctrl+K in main.rs:299-313: pretty JSON, gzip when the extension is .gz."""
        _check(lib.bt_scene_save(self._h, os.fspath(path).encode()))

    def __del__(self):
        h = getattr(self, "_h", None)
        if h:
            lib.bt_scene_free(h)
            self._h = None

    def find_by_tag(self, tag) -> Optional[int]:
        """Scene::find_by_tag (scene/mod.rs:124-129): ObjectRef or None."""
        ref = C.c_uint64()
        rc = lib.bt_scene_find_by_tag(self._h, tag.encode(), C.byref(ref))
        return ref.value if rc == 0 else None

    def set_camera_aspect(self, camera_ref, aspect_ratio):
        """main.rs:218-223."""
        _check(lib.bt_scene_set_camera_aspect(self._h, camera_ref, aspect_ratio))

    def set_lens(self, centre, rs, step, radius, max_steps=4096):
        """EXTENSION, not in the reference (include/bendy_hip.h `bt_lens`): bend rays around a point mass."""
        lens = _CLens((C.c_float * 3)(*[float(v) for v in centre]), rs, step, radius, max_steps)
        _check(lib.bt_scene_set_lens(self._h, C.byref(lens)))

    def clear_lens(self):
        _check(lib.bt_scene_set_lens(self._h, None))

    @property
    def object_count(self):
        return lib.bt_scene_object_count(self._h)

    @property
    def data_count(self):
        return lib.bt_scene_data_count(self._h)

    def export_prims(self):
        """The flattened primitive table exactly as uploaded to the GPU ([n, 36] float32 view)."""
        n = _check(lib.bt_scene_export_prims(self._h, None, 0))
        out = np.zeros(n, dtype=np.float32)
        _check(lib.bt_scene_export_prims(self._h, out.ctypes.data_as(C.POINTER(C.c_float)), n))
        return out.reshape(-1, 36)

    def set_tuning(self, **knobs):
        """bt_scene_set_tuning: pins launch-shape knobs of this handle (tests and A/B tools; none of them changes a
        pixel).  Keywords = fields of `bt_tuning` (slices, phase_vote, scratch_cap_bytes, packed);
        fields not named keep their current value; no keywords = defaults."""
        t = _CTuning()
        if not knobs:
            _check(lib.bt_scene_set_tuning(self._h, None))
            return
        _check(lib.bt_scene_get_tuning(self._h, C.byref(t)))
        for k, v in knobs.items():
            if not hasattr(t, k):
                raise TypeError(f"bt_tuning has no field {k!r}")
            setattr(t, k, int(v))
        _check(lib.bt_scene_set_tuning(self._h, C.byref(t)))

    def tuning(self) -> dict:
        t = _CTuning()
        _check(lib.bt_scene_get_tuning(self._h, C.byref(t)))
        return {k: getattr(t, k) for k, _ in _CTuning._fields_}

    def tuning_from_env(self, environ=None):
        """Developer convenience for tools/ and tests/: BT_SLICES, BT_PHASE_VOTE, BT_SCRATCH_CAP, BT_PACKED -> set_tuning().  The library itself never reads the environment."""
        env = os.environ if environ is None else environ
        names = {"BT_SLICES": "slices", "BT_PHASE_VOTE": "phase_vote", "BT_SCRATCH_CAP": "scratch_cap_bytes", "BT_PACKED": "packed"}
        knobs = {f: int(env[e]) for e, f in names.items() if env.get(e) not in (None, "", "-")}
        if knobs:
            self.set_tuning(**knobs)
        return knobs

    def trim(self):
        """Returns the scratch / cached frame the handle keeps between calls to the device (bt_scene_trim)."""
        _check(lib.bt_scene_trim(self._h))

    def last_stats(self) -> Stats:
        st = Stats()
        _check(lib.bt_scene_last_stats(self._h, C.byref(st)))
        return st


class Buffer:
    """`Buffer` (tracer/buffer.rs:32-179): RGBA32F running sums + sample counter.

    device="cuda" keeps the sums in HBM (a torch tensor, plumbing only); device="cpu" keeps
    a numpy array and every render call copies it through the device."""

    def __init__(self, width, height, color_space=ColorSpace.SRgb, device="cuda"):
        self.width, self.height = int(width), int(height)
        self.color_space = ColorSpace(color_space)
        self.samples = 0
        self.device = device
        if device == "cpu":
            self.data = np.zeros((self.height, self.width, 4), dtype=np.float32)
            self.data[..., 3] = 1.0  # BLACK_ALPHA_ONE, buffer.rs:9,43
        else:
            import torch
            self.data = torch.zeros((self.height, self.width, 4), dtype=torch.float32, device=device)
            self.data[..., 3] = 1.0

    @classmethod
    def new(cls, width, height, color_space=ColorSpace.SRgb, device="cuda"):
        return cls(width, height, color_space, device)

    def dimensions(self):
        return (self.width, self.height)

    def pixel_width(self):  # buffer.rs:68-71
        return float(np.float32(2.0) * (np.float32(1.0) / np.float32(self.width)))

    def pixel_height(self):  # buffer.rs:73-76
        return float(np.float32(2.0) * (np.float32(1.0) / np.float32(self.height)))

    def clear(self):  # buffer.rs:82-87
        self.data[...] = 0.0
        self.data[..., 3] = 1.0
        self.samples = 0

    def inc_samples(self, n):  # buffer.rs:155-157
        self.samples += n

    def chunks(self, chunks_x, chunks_y):
        """Buffer::chunks (buffer.rs:102-115, 293-326): list of (min_x, min_y, max_x, max_y)."""
        cw = self.width // chunks_x + (1 if self.width % chunks_x else 0)
        ch = self.height // chunks_y + (1 if self.height % chunks_y else 0)
        out, oy = [], 0
        while oy < self.height:
            h = min(ch, self.height - oy)
            ox = 0
            while ox < self.width:
                w = min(cw, self.width - ox)
                out.append((ox, oy, ox + w, oy + h))
                ox += w
            oy += h
        return out

    def numpy(self):
        return self.data if self.device == "cpu" else self.data.cpu().numpy()

    def mean(self):
        """sum / samples as buffer.rs:124-127 does before the colour-space conversion."""
        return self.numpy()[..., :3] / np.float32(max(self.samples, 1))

    def preview(self):
        """Buffer::preview (buffer.rs:117-138) -> uint8 [H, W, 4] on the host."""
        if self.device == "cpu":
            out = np.zeros((self.height, self.width, 4), dtype=np.uint8)
            _check(lib.bt_preview(self.data.ctypes.data_as(C.POINTER(C.c_float)),
                                  out.ctypes.data_as(C.POINTER(C.c_uint8)), self.width, self.height,
                                  max(self.samples, 1), int(self.color_space)))
            return out
        import torch
        out = torch.empty((self.height, self.width, 4), dtype=torch.uint8, device=self.data.device)
        _check(lib.bt_preview_device(self.data.data_ptr(), out.data_ptr(), self.width, self.height,
                                     max(self.samples, 1), int(self.color_space),
                                     torch.cuda.current_stream().cuda_stream))
        return out.cpu().numpy()


class Tracer:
    """`Tracer` (tracer/mod.rs:165-203)."""

    DEFAULT_SEED = 0x5EED

    def __init__(self, config: Optional[Config] = None):
        self.config = config or Config()

    @classmethod
    def new(cls):
        return cls()

    @classmethod
    def with_config(cls, config):
        return cls(config)

    def render(self, scene: Scene, camera: int, config: RenderConfig, buffer: Buffer, seed: Optional[int] = None,
               sample_base: Optional[int] = None) -> Status:
        """Tracer::render (mod.rs:179-202).  `seed` stands in for SmallRng::from_entropy()
        (mod.rs:239-242); `sample_base` defaults to buffer.samples / n^2 so that successive
        calls on one buffer draw fresh samples from the same seed."""
        seed = self.DEFAULT_SEED if seed is None else seed
        nn = config.subsample.subpixel_count()
        if sample_base is None:
            # the next unused sample index: ceil, so that a change of `subsample` between calls never replays indices
            sample_base = (buffer.samples + nn - 1) // nn
        c, r = _c_configs(self.config, config, sample_base)
        if buffer.device == "cpu":
            rc = lib.bt_render(scene._h, camera, C.byref(c), C.byref(r),
                               buffer.data.ctypes.data_as(C.POINTER(C.c_float)), buffer.width, buffer.height, seed)
        else:
            import torch
            rc = lib.bt_render_device(scene._h, camera, C.byref(c), C.byref(r), buffer.data.data_ptr(), buffer.width,
                                      buffer.height, seed, torch.cuda.current_stream().cuda_stream)
        _check(rc)
        if rc == Status.InProgress:
            buffer.inc_samples(config.samples * nn)  # mod.rs:199
        return Status(rc)

    # ---- multi-GPU tile sharding (DESIGN.md "Multi-GPU") ----
    def render_shard(self, scene: Scene, camera: int, config: RenderConfig, shard, width, height, rank, world,
                     seed: Optional[int] = None, sample_base: int = 0) -> Status:
        import torch
        seed = self.DEFAULT_SEED if seed is None else seed
        c, r = _c_configs(self.config, config, sample_base)
        assert shard.numel() == shard_floats(width, height, world) and shard.dtype == torch.float32
        rc = lib.bt_render_shard_device(scene._h, camera, C.byref(c), C.byref(r), shard.data_ptr(), width, height, rank,
                                        world, seed, torch.cuda.current_stream().cuda_stream)
        return Status(_check(rc))


class Comm:
    """`bt_comm` (include/bendy_hip.h): the RCCL communicator of the frame exchange, behind the C ABI -- what a host
    without an RCCL binding of its own calls.  One process per GPU; `unique_id()` on rank 0, its 128 bytes handed to
    the other ranks by any host-side channel, then `Comm(rank, world, uid)` on every rank (collective)."""

    ID_BYTES = 128

    @staticmethod
    def unique_id() -> bytes:
        buf = C.create_string_buffer(Comm.ID_BYTES)
        _check(lib.bt_comm_unique_id(buf, Comm.ID_BYTES))
        return buf.raw

    def __init__(self, rank, world, uid: bytes):
        assert len(uid) == self.ID_BYTES
        h = lib.bt_comm_init(rank, world, uid, len(uid))
        if not h:
            raise BendyError(lib.bt_last_error_code(), lib.bt_last_error().decode("utf-8", "replace"))
        self._h = C.c_void_p(h)
        self.rank, self.world = rank, world

    def close(self):
        if getattr(self, "_h", None):
            lib.bt_comm_free(self._h)
            self._h = None

    __del__ = close

    def allgather(self, shard, gathered, width, height):
        import torch
        assert gathered.numel() == self.world * shard.numel() == self.world * shard_floats(width, height, self.world)
        _check(lib.bt_allgather_shards_device(self._h, shard.data_ptr(), gathered.data_ptr(), width, height,
                                              torch.cuda.current_stream().cuda_stream))

    def exchange(self, shard, gathered, buffer: "Buffer"):
        """All-gather + un-permute: every rank's `buffer` then holds the whole frame of running sums."""
        import torch
        _check(lib.bt_exchange_frame_device(self._h, shard.data_ptr(), gathered.data_ptr(), buffer.data.data_ptr(),
                                            buffer.width, buffer.height, torch.cuda.current_stream().cuda_stream))


def write_png(path, rgba8):
    """buffer.preview().save(path) (main.rs:275-298)."""
    a = np.ascontiguousarray(rgba8, dtype=np.uint8)
    _check(lib.bt_write_png(os.fspath(path).encode(), a.ctypes.data_as(C.POINTER(C.c_uint8)), a.shape[1], a.shape[0]))


def shard_floats(width, height, world):
    return lib.bt_shard_floats(width, height, world)


def new_shard(width, height, world, device="cuda"):
    """A fresh shard accumulator: zeros with alpha = 1 (Buffer::new, buffer.rs:41-50)."""
    import torch
    s = torch.zeros(shard_floats(width, height, world), dtype=torch.float32, device=device)
    s.view(-1, 4)[:, 3] = 1.0
    return s


def unshard(gathered, buffer: Buffer, world):
    import torch
    _check(lib.bt_unshard_device(gathered.data_ptr(), buffer.data.data_ptr(), buffer.width, buffer.height, world,
                                 torch.cuda.current_stream().cuda_stream))


def tile_owner_map(width, height, world):
    """Which rank owns each BT_TILE x BT_TILE tile: tile t (row-major) -> t % world."""
    tx, ty = (width + BT_TILE - 1) // BT_TILE, (height + BT_TILE - 1) // BT_TILE
    return (np.arange(tx * ty) % world).reshape(ty, tx)
