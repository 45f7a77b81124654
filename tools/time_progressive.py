"""Times the reference's interactive pattern (main.rs:245-254): one Tracer::render call per frame with samples = 1,
Subpixel(2) (4 rays per pixel) on a 1080p frame, 50 calls (developer tool).  bt_tuning pins the pixel-block split."""
import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..'))
import torch
import bendy_tracer_amd as b
w, h = 1920, 1080
for name in ('scene', 'cornell2', 'volume'):
    sc = b.Scene.load(f'scenes/{name}.json.gz'); cam = sc.find_by_tag('camera'); sc.set_camera_aspect(cam, w / h)
    tr = b.Tracer.with_config(b.Config(chunks_x=8, chunks_y=4))
    for mode in ('auto', 't1', 't2', 't4', 'lanes'):   # tN = N whole tiles per workgroup; lanes = no work queue
        sc.set_tuning()
        if mode == 'lanes':
            sc.set_tuning(queue=0)
        elif mode != 'auto':
            sc.set_tuning(tiles_per_wg=int(mode[1:]))
        buf = b.Buffer.new(w, h)
        rc = b.RenderConfig.with_samples_subsample(1, b.Subsample(2))
        for i in range(5):
            tr.render(sc, cam, rc, buf)
        torch.cuda.synchronize()
        t = time.perf_counter()
        n = 50
        for i in range(n):
            tr.render(sc, cam, rc, buf)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t) / n
        st = sc.last_stats()
        print(f'{name:9s} slices={mode:>4} ({st.slices}): {dt*1e3:7.3f} ms per call (kernel {st.kernel_ms:.3f} ms), {w*h*4/dt/1e6:9.1f} Msamples/s', flush=True)
