"""Developer tool: one ray per pixel per call (T = 1) and T = 2 through the block queue with 1 / 2 / 4 tiles per workgroup
vs the lanes kernel, 1080p."""
import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..'))
import torch
import bendy_tracer_amd as b
w, h = 1920, 1080
for name in ('scene', 'cornell2', 'volume'):
    sc = b.Scene.load(f'scenes/{name}.json.gz'); cam = sc.find_by_tag('camera'); sc.set_camera_aspect(cam, w / h)
    tr = b.Tracer.with_config(b.Config(chunks_x=8, chunks_y=4))
    for samples in (1, 2, 3):
        row = []
        for mode in ('auto', 'lanes', 'q1t1', 'q1t2', 'q1t4'):
            sc.set_tuning()
            if mode == 'lanes':
                sc.set_tuning(queue=0)
            elif mode.startswith('q1'):
                sc.set_tuning(queue=1, slices=1, tiles_per_wg=int(mode[3:]))
            buf = b.Buffer.new(w, h)
            rc = b.RenderConfig.with_samples(samples)
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
            row.append(f'{mode}:{dt*1e3:.3f}')
        print(f'{name:9s} T={samples} ms per call  ' + '  '.join(row), flush=True)
