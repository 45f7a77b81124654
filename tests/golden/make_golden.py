"""Generates tests/golden/*.npz with the CPU oracle (oracle/bt_oracle.c).

SELF-GENERATED fixtures: the reference (Rust) cannot be built or run in this image and seeds
its RNG from OS entropy, so these are NOT outputs of the reference binary.  They pin the
oracle (and through it the numerics contract) against accidental change, and give the GPU
tests committed expected framebuffers.  Usage: python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import bt_oracle_py as o  # noqa: E402

SEED = 0x5EED
# name -> (scene, width, height, samples, subsample_n, output)
CASES = {
    "scene_64x36_s4": ("scene", 64, 36, 4, 0, 0),
    "scene_64x36_s1_n2": ("scene", 64, 36, 1, 2, 0),
    "cornell_48x48_s4": ("cornell", 48, 48, 4, 0, 0),
    "cornell2_48x48_s4": ("cornell2", 48, 48, 4, 0, 0),
    "volume_60x40_s4": ("volume", 60, 40, 4, 0, 0),
    "cloud_60x40_s4": ("cloud", 60, 40, 4, 0, 0),
    "scene_64x36_albedo": ("scene", 64, 36, 2, 0, 1),
    "scene_64x36_normal": ("scene", 64, 36, 2, 0, 2),
    "volume_60x40_depth": ("volume", 60, 40, 2, 0, 3),
}


def render_case(case, recursive):
    name, w, h, spp, n, out = CASES[case]
    sc = o.Scene.load(os.path.join(ROOT, "scenes", f"{name}.json.gz"))
    cam = sc.find_by_tag("camera")
    sc.set_camera_aspect(cam, w / h)
    img, rc, seg = o.render(sc, cam, o.default_config(samples=spp, subsample_n=n, output=out, recursive=recursive), w, h,
                            SEED, nthreads=8)
    return img, seg


if __name__ == "__main__":
    for case in CASES:
        it, seg = render_case(case, 0)
        rec, seg2 = render_case(case, 1)
        assert seg == seg2
        np.savez_compressed(os.path.join(HERE, case + ".npz"), iterative=it, recursive=rec, segments=np.uint64(seg))
        print(case, it.shape, seg)
