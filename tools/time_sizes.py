"""Throughput vs samples per pixel at 1080p (developer tool): more samples per lane = better intra-wave balance."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..'))
import torch
import bendy_tracer_amd as b
for name in ('scene', 'cornell'):
    for spp in (4, 16, 64, 256, 1024):
        w, h = 1920, 1080
        gs = b.Scene.load(f'scenes/{name}.json.gz'); cam = gs.find_by_tag('camera'); gs.set_camera_aspect(cam, w / h); gs.tuning_from_env()
        tr = b.Tracer.with_config(b.Config(chunks_x=8, chunks_y=4)); buf = b.Buffer.new(w, h); ks=[]
        for it in range(4):
            tr.render(gs, cam, b.RenderConfig.with_samples(spp), buf, sample_base=it * spp); st=gs.last_stats(); ks.append(st.kernel_ms)
        k=min(ks[1:]); print(f'{name:9s} {w}x{h}x{spp}: kernel {k:9.3f} ms  {w*h*spp/k/1e3:9.1f} Msamples/s', flush=True)
