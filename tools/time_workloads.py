"""Times the BASELINE workloads on the GPU (developer tool).  BT_SLICES / BT_QUEUE / BT_PHASE_VOTE / ... in the
environment are turned into bt_tuning fields here (Scene.tuning_from_env); the library does not read them."""
import sys, time, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..'))
import torch
import bendy_tracer_amd as b
ONLY = os.environ.get('BT_ONLY', '').split(',') if os.environ.get('BT_ONLY') else None
W = [('scene', 1920, 1080, 64), ('cornell2', 512, 512, 16), ('volume', 1920, 1080, 64), ('cornell', 1920, 1080, 64), ('cloud', 1920, 1080, 64)]
for name, w, h, spp in W:
    if ONLY and name not in ONLY:
        continue
    gs = b.Scene.load(f'scenes/{name}.json.gz'); cam = gs.find_by_tag('camera'); gs.set_camera_aspect(cam, w / h); gs.tuning_from_env()
    tr = b.Tracer.with_config(b.Config(chunks_x=8, chunks_y=4))
    buf = b.Buffer.new(w, h)
    ks = []
    for it in range(6):
        tr.render(gs, cam, b.RenderConfig.with_samples(spp), buf, sample_base=it * spp)
        ks.append(gs.last_stats().kernel_ms)
    st = gs.last_stats()
    k = min(ks[1:])
    print(f'{name:9s} {w}x{h}x{spp}: kernel {k:8.3f} ms  {w*h*spp/k/1e3:9.1f} Msamples/s  seg/sample {st.segments/st.samples:.3f}', flush=True)
