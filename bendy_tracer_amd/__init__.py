"""bendy_tracer_amd -- MI355X (gfx950) implementation of bendy-tracer's per-sample hot path.

Only what the path needs lives here: `csrc/` (HIP kernels + scene loader + C ABI, built
into `libbendy_hip.so`) and `api.py` (host-side mirror of the reference's Rust API).
Importing the package loads the shared library; a missing library is an ImportError.
"""
from .api import (BendyError, Buffer, ColorSpace, Comm, Config, Output, RenderConfig, Scene, Stats, Status, Subsample,
                  Tracer, new_shard, shard_floats, tile_owner_map, unshard, write_png)

__all__ = ["BendyError", "Buffer", "ColorSpace", "Comm", "Config", "Output", "RenderConfig", "Scene", "Stats", "Status",
           "Subsample", "Tracer", "new_shard", "shard_floats", "tile_owner_map", "unshard", "write_png"]
