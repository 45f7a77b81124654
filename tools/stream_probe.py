"""First contact of the streaming queue with the GPU: small frames against the oracle, each under its own process
timeout (developer tool; a hang here must not take the box)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..')); sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..', 'oracle')); sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..', 'tests'))
import numpy as np, torch
import bendy_tracer_amd as b
import bt_oracle_py as o
from helpers import gpu_render, oracle_render
case = sys.argv[1].split(',')
name, w, h, spp = case[0], int(case[1]), int(case[2]), int(case[3])
tuning = {}
for kv in case[4:]:
    k, v = kv.split('='); tuning[k] = int(v)
buf, st, _ = gpu_render(b, name, w, h, spp, tuning=tuning or None)
it, seg = oracle_render(o, name, w, h, spp, threads=8)
got = buf.numpy()
same = np.array_equal(got, it)
if not same:
    bad = np.argwhere((got != it).any(axis=-1))
    print(f"  {len(bad)} of {w*h} pixels differ; first {bad[:6].tolist()}; tile rows {sorted(set((bad[:,0]//16).tolist()))[:12]} cols {sorted(set((bad[:,1]//16).tolist()))[:12]}")
    y, x = bad[0]; print("   got", got[y, x], "want", it[y, x], "ratio", got[y, x, :3] / np.maximum(it[y, x, :3], 1e-9))
print(f"debug raw {st.lens_steps:#x}: code {st.lens_steps >> 60}, a {(st.lens_steps >> 32) & 0xfffffff}, b {(st.lens_steps >> 16) & 0xffff}, c {(st.lens_steps >> 8) & 0xff}, d {st.lens_steps & 0xff}")
print(f"debug: units summed {st.lens_steps >> 40}, sum of n_act {(st.lens_steps >> 20) & 0xfffff}, pixels summed {st.lens_steps & 0xfffff}")
print(f"{sys.argv[1]}: slices {st.slices} launches {st.launches} scratch {st.scratch_bytes} parked {st.parked_bytes} kernel {st.kernel_ms:.3f} ms  segments {st.segments} vs {seg}  bit-exact {same}", flush=True)
sys.exit(0 if same and st.segments == seg else 1)
