"""Developer tool: shallow launches (T = 1, 2, 4, 8 rays per pixel per call) on a 1080p frame under tiles-per-workgroup /
slices / lanes settings -- the data behind bt_api.cpp's choice for the interactive pattern (main.rs:245-254)."""
import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..'))
import torch
import bendy_tracer_amd as b
w, h = (int(x) for x in os.environ.get('BT_FRAME', '1920x1080').split('x'))
names = os.environ.get('BT_ONLY', 'scene,cornell2,volume').split(',')
for name in names:
    sc = b.Scene.load(f'scenes/{name}.json.gz'); cam = sc.find_by_tag('camera'); sc.set_camera_aspect(cam, w / h)
    tr = b.Tracer.with_config(b.Config(chunks_x=8, chunks_y=4))
    for samples, sub in ((1, 1), (2, 1), (1, 2), (8, 1), (16, 1), (32, 1), (128, 1)):
        row = []
        for mode in ('auto', 't1', 't2', 's2', 's4', 's8', 's16', 'lanes'):
            sc.set_tuning()
            if mode == 'lanes':
                sc.set_tuning(queue=0)
            elif mode[0] == 't':
                sc.set_tuning(tiles_per_wg=int(mode[1:]), slices=1)
            elif mode[0] == 's':
                sc.set_tuning(slices=int(mode[1:]))
            buf = b.Buffer.new(w, h)
            rc = b.RenderConfig.with_samples_subsample(samples, b.Subsample(sub))
            try:
                for i in range(3):
                    tr.render(sc, cam, rc, buf)
                torch.cuda.synchronize()
                t = time.perf_counter()
                n = 30
                for i in range(n):
                    tr.render(sc, cam, rc, buf)
                torch.cuda.synchronize()
                dt = (time.perf_counter() - t) / n
                st = sc.last_stats()
                row.append(f'{mode}:{dt*1e3:.3f}({st.slices})')
            except Exception as e:
                row.append(f'{mode}:err')
        print(f'{name:9s} T={samples*sub*sub:3d} ms per call  ' + '  '.join(row), flush=True)
