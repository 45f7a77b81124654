"""Random scene.json documents for fuzz-style parity tests (test infrastructure).

Same on-disk schema as the bundled files: spheres (plain, volumetric, light),
rects, cuboids built the way Cuboid::new builds them (cuboid.rs:19-30), all five materials,
density maps, rotated / translated / mildly scaled transforms, camera with or without focus.
"""
import json
import math

import numpy as np


def _rot(rng):
    a, b, c = rng.uniform(-math.pi, math.pi, 3)
    ry = np.array([[math.cos(a), 0, math.sin(a)], [0, 1, 0], [-math.sin(a), 0, math.cos(a)]])
    rx = np.array([[1, 0, 0], [0, math.cos(b), -math.sin(b)], [0, math.sin(b), math.cos(b)]])
    rz = np.array([[math.cos(c), -math.sin(c), 0], [math.sin(c), math.cos(c), 0], [0, 0, 1]])
    return ry @ rx @ rz


def _affine(m, t):
    m = np.asarray(m, dtype=np.float32)
    return [float(v) for v in np.concatenate([m[:, 0], m[:, 1], m[:, 2], np.asarray(t, np.float32)])]


def _rect(material, x, y):
    """Rect::new (rect.rs:22-36)."""
    x, y = np.asarray(x, np.float32), np.asarray(y, np.float32)
    hw, hh = np.float32(np.linalg.norm(x)), np.float32(np.linalg.norm(y))
    xn, yn = x / hw, y / hh
    z = np.cross(xn, yn).astype(np.float32)
    return {"material": material, "half_width": float(hw), "half_height": float(hh),
            "x": [float(v) for v in xn], "y": [float(v) for v in yn], "z": [float(v) for v in z]}


def _cuboid(material, hx, hy, hz):
    """Cuboid::new (cuboid.rs:19-30)."""
    x, y, z = np.array([hx, 0, 0.0]), np.array([0, hy, 0.0]), np.array([0, 0, hz])
    faces = [(-z, _rect(material, x, y)), (z, _rect(material, -x, y)), (-x, _rect(material, z, y)),
             (x, _rect(material, -z, y)), (-y, _rect(material, x, z)), (y, _rect(material, x, -z))]
    return {"faces": [[[float(v) for v in off], r] for off, r in faces]}


def random_scene(seed, n_objects=8, volume_prob=0.25, focus_prob=0.5, scale_prob=0.3, n_lights=(1, 3), density_dims=(3, 5, 8)):
    rng = np.random.default_rng(seed)
    col = lambda lo=0.1, hi=0.95: dict(zip("rgb", [float(v) for v in rng.uniform(lo, hi, 3)]))
    data = {"0": {"inner": {"Material": {"Flat": {"albedo": {"r": 0.0, "g": 0.0, "b": 0.0}}}}}}

    def add_data(inner):
        k = str(len(data))
        data[k] = {"inner": inner}
        return int(k)

    root_kind = rng.integers(3)
    if root_kind == 0:
        root = add_data({"Material": {"Emissive": {"albedo": col(), "intensity": float(rng.uniform(0.05, 0.5))}}})
    elif root_kind == 1:
        root = add_data({"Material": {"Flat": {"albedo": col(0.0, 0.3)}}})
    else:
        root = 0
    mats = [
        add_data({"Material": {"Diffuse": {"albedo": col(), "roughness": 0.5}}}),
        add_data({"Material": {"Diffuse": {"albedo": col(), "roughness": 1.0}}}),
        add_data({"Material": {"Metallic": {"albedo": col(), "roughness": float(rng.uniform(0.0, 0.4))}}}),
        add_data({"Material": {"Glass": {"albedo": col(0.8, 1.0), "roughness": float(rng.uniform(0.0, 0.1)),
                                         "ior": float(rng.uniform(1.1, 1.8))}}}),
        add_data({"Material": {"Flat": {"albedo": col()}}}),
    ]
    light_mat = add_data({"Material": {"Emissive": {"albedo": col(0.7, 1.0), "intensity": float(rng.uniform(5, 20))}}})

    def density_map():
        n = int(rng.choice(list(density_dims)))
        buf = rng.uniform(0, 1, n * n * n).astype(np.float32)
        buf[rng.uniform(size=buf.size) < 0.5] = 0.0
        buf *= np.float32(rng.choice([0.5, 3.0, 12.0]))      # up to density*step >= 1
        return add_data({"Volume": {"DensityMap": {"width": n, "height": n, "depth": n,
                                                   "size": [n - 1.0] * 3, "buffer": [float(v) for v in buf]}}})

    objects = {}

    def add_obj(inner, m, t, tag=None, flags=0):
        k = len(objects)
        a = _affine(m, t)
        objects[str(k)] = {"object_ref": k, "tag": tag, "flags": {"bits": flags},
                           "transform": {"transform_world": a, "transform_local": a, "transform_parent": None},
                           "inner": inner, "children": None}

    # camera looking down -z from z = 9, small random tilt
    tilt = _rot(np.random.default_rng(seed + 1)) if False else np.eye(3)
    a = rng.uniform(-0.15, 0.15)
    tilt = np.array([[math.cos(a), 0, math.sin(a)], [0, 1, 0], [-math.sin(a), 0, math.cos(a)]])
    focus = float(rng.uniform(6, 12)) if rng.uniform() < focus_prob else None
    add_obj({"Camera": {"sensor_size": 0.024, "focal_length": float(rng.uniform(0.03, 0.06)), "aspect_ratio": 1.5,
                        "fstop": float(rng.uniform(0.7, 4.0)), "focus": focus}}, tilt, (0, 0.5, 9), tag="camera")
    # ground so that most paths do something
    add_obj({"Sphere": {"material": mats[0], "volume": None, "radius": 100.0}}, np.eye(3), (0, -102.0, 0))
    lights = int(rng.integers(n_lights[0], n_lights[1] + 1))
    for i in range(n_objects):
        is_light = i < lights
        mat = light_mat if is_light else int(rng.choice(mats))
        pos = rng.uniform([-3.5, -1.5, -5], [3.5, 2.5, 3])
        if is_light:
            pos[1] = rng.uniform(2.5, 5)
        m = _rot(rng)
        if rng.uniform() < scale_prob:
            m = m @ np.diag(rng.uniform(0.7, 1.4, 3))
        kind = rng.integers(3)
        flags = 1 if is_light else 0
        if kind == 0:
            vol = density_map() if (not is_light and rng.uniform() < volume_prob) else None
            add_obj({"Sphere": {"material": mat, "volume": vol, "radius": float(rng.uniform(0.3, 1.2))}}, m, pos, flags=flags)
        elif kind == 1:
            x = np.array([rng.uniform(0.3, 1.5), 0, 0]); y = np.array([0, rng.uniform(0.3, 1.5), 0])
            add_obj({"Rect": _rect(mat, x, y)}, m, pos, flags=flags)
        else:
            add_obj({"Cuboid": _cuboid(mat, *rng.uniform(0.2, 0.9, 3))}, m, pos, flags=flags)
    return json.dumps({"roots": [], "root_material": root,
                       "objects": {"collection": objects, "next_key": len(objects)},
                       "data": {"collection": data, "next_key": len(data)}})
