"""Developer tool: kernel time of one BASELINE workload over many launches (min / median / mean of the library's HIP-event times).
usage: BT_ONLY=scene python tools/time_c3.py [launches]"""
import sys, os, statistics
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..'))
import torch
import bendy_tracer_amd as b
n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
W = {'scene': (1920, 1080, 64), 'cornell2': (512, 512, 16), 'volume': (1920, 1080, 64), 'cornell': (1920, 1080, 64), 'cloud': (1920, 1080, 64)}
for name in os.environ.get('BT_ONLY', 'scene').split(','):
    w, h, spp = W[name]
    gs = b.Scene.load(f'scenes/{name}.json.gz'); cam = gs.find_by_tag('camera'); gs.set_camera_aspect(cam, w / h); gs.tuning_from_env()
    tr = b.Tracer.with_config(b.Config(chunks_x=8, chunks_y=4))
    buf = b.Buffer.new(w, h)
    ks = []
    for it in range(n + 5):
        tr.render(gs, cam, b.RenderConfig.with_samples(spp), buf, sample_base=(it % 8) * spp)
        ks.append(gs.last_stats().kernel_ms)
    ks = ks[5:]
    print(f'{name:9s} {w}x{h}x{spp}: kernel min {min(ks):.3f}  median {statistics.median(ks):.3f}  mean {statistics.mean(ks):.3f} ms over {n} launches', flush=True)
