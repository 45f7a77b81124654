"""Shared test helpers (test infrastructure)."""
import json

import numpy as np

from conftest import scene_path

BT_TILE = 16


def oracle_scene(o, name, w, h):
    sc = o.Scene.load(scene_path(name))
    cam = sc.find_by_tag("camera")
    sc.set_camera_aspect(cam, w / h)   # main.rs:218-223
    return sc, cam


def oracle_render(o, name, w, h, spp, n=0, output=0, recursive=0, seed=0x5EED, threads=8, sample_base=0, **kw):
    sc, cam = oracle_scene(o, name, w, h)
    cfg = o.default_config(samples=spp, subsample_n=n, output=output, recursive=recursive, sample_base=sample_base, **kw)
    img, rc, seg = o.render(sc, cam, cfg, w, h, seed, nthreads=threads)
    return img, seg


def gpu_scene(b, name, w, h, tuning=None):
    """`tuning`: bt_tuning fields (Scene.set_tuning) that pin the launch shape for this handle."""
    sc = b.Scene.load(scene_path(name))
    cam = sc.find_by_tag("camera")
    sc.set_camera_aspect(cam, w / h)
    if tuning:
        sc.set_tuning(**tuning)
    return sc, cam


def gpu_render(b, name, w, h, spp, n=0, output=0, seed=0x5EED, device="cuda", tuning=None, **rc_kw):
    import torch
    sc, cam = gpu_scene(b, name, w, h, tuning)
    buf = b.Buffer.new(w, h, device=device)
    tr = b.Tracer.with_config(b.Config(chunks_x=8, chunks_y=4, output=b.Output(output)))
    st = tr.render(sc, cam, b.RenderConfig(samples=spp, subsample=b.Subsample(n), **rc_kw), buf, seed=seed)
    if device != "cpu":
        torch.cuda.synchronize()
    return buf, sc.last_stats(), st


def unshard_numpy(gathered, w, h, world):
    """Host mirror of bt_unshard_device: `world` shards back to back -> row-major frame."""
    tx, ty = (w + BT_TILE - 1) // BT_TILE, (h + BT_TILE - 1) // BT_TILE
    per_rank = (tx * ty + world - 1) // world
    g = np.asarray(gathered, dtype=np.float32).reshape(world, per_rank, BT_TILE, BT_TILE, 4)
    frame = np.zeros((h, w, 4), dtype=np.float32)
    for t in range(tx * ty):
        r, slot = t % world, t // world
        x0, y0 = (t % tx) * BT_TILE, (t // tx) * BT_TILE
        ww, hh = min(BT_TILE, w - x0), min(BT_TILE, h - y0)
        frame[y0:y0 + hh, x0:x0 + ww] = g[r, slot, :hh, :ww]
    return frame


def shard_from_frame(frame, rank, world):
    """Inverse of unshard_numpy for one rank (pads with zeros / alpha 1)."""
    h, w, _ = frame.shape
    tx, ty = (w + BT_TILE - 1) // BT_TILE, (h + BT_TILE - 1) // BT_TILE
    per_rank = (tx * ty + world - 1) // world
    s = np.zeros((per_rank, BT_TILE, BT_TILE, 4), dtype=np.float32)
    s[..., 3] = 1.0
    for t in range(rank, tx * ty, world):
        x0, y0 = (t % tx) * BT_TILE, (t // tx) * BT_TILE
        ww, hh = min(BT_TILE, w - x0), min(BT_TILE, h - y0)
        s[t // world, :hh, :ww] = frame[y0:y0 + hh, x0:x0 + ww]
    return s.reshape(-1)


def flat_scene_json(sphere_color=(0.25, 0.5, 0.75), root_color=(1.0, 1.0, 1.0), root_intensity=0.5,
                    focus=None, extra_objects=None, extra_data=None):
    """A scene whose materials are only Flat / Emissive: no stochastic shading, so with
    focus=None interior pixels are exact constants (KAT 9)."""
    ident = [1, 0, 0, 0, 1, 0, 0, 0, 1]
    def obj(ref, inner, t, tag=None, flags=0):
        return {"object_ref": ref, "tag": tag, "flags": {"bits": flags},
                "transform": {"transform_world": ident + list(t), "transform_local": ident + list(t),
                              "transform_parent": None}, "inner": inner, "children": None}
    objects = {
        "0": obj(0, {"Camera": {"sensor_size": 0.024, "focal_length": 0.05, "aspect_ratio": 1.0, "fstop": 2.0,
                                "focus": focus}}, (0, 0, 5), tag="camera"),
        "1": obj(1, {"Sphere": {"material": 2, "volume": None, "radius": 1.0}}, (0, 0, 0)),
    }
    data = {
        "0": {"inner": {"Material": {"Flat": {"albedo": {"r": 0.0, "g": 0.0, "b": 0.0}}}}},
        "1": {"inner": {"Material": {"Emissive": {"albedo": dict(zip("rgb", root_color)), "intensity": root_intensity}}}},
        "2": {"inner": {"Material": {"Flat": {"albedo": dict(zip("rgb", sphere_color))}}}},
    }
    if extra_objects:
        objects.update(extra_objects)
    if extra_data:
        data.update(extra_data)
    doc = {"roots": [], "root_material": 1,
           "objects": {"collection": objects, "next_key": len(objects)},
           "data": {"collection": data, "next_key": len(data)}}
    return json.dumps(doc)
