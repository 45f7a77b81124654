"""Developer tool: C5 (scene.json 3840x2160x256 spp) on one GPU under several scratch caps (launches per render)."""
import sys, os, statistics
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..'))
import torch
import bendy_tracer_amd as b
w, h, spp = 3840, 2160, 256
for cap_gib in (2, 4, 8, 16, 32):
    gs = b.Scene.load('scenes/scene.json.gz'); cam = gs.find_by_tag('camera'); gs.set_camera_aspect(cam, w / h)
    gs.set_tuning(scratch_cap_bytes=cap_gib << 30)
    tr = b.Tracer.with_config(b.Config(chunks_x=8, chunks_y=4))
    buf = b.Buffer.new(w, h)
    ks = []
    for it in range(6):
        tr.render(gs, cam, b.RenderConfig.with_samples(spp), buf, sample_base=it * spp)
        ks.append(gs.last_stats().kernel_ms)
    st = gs.last_stats()
    print(f'cap {cap_gib:2d} GiB: launches {st.launches:2d} slices {st.slices} kernel min {min(ks[1:]):.3f} median {statistics.median(ks[1:]):.3f} ms  scratch {st.scratch_bytes/2**30:.2f} GiB', flush=True)
    gs.trim()
