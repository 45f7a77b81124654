"""Exploratory GPU-vs-oracle probe (developer tool, not part of the product)."""
import sys, time, os
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..'))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..', 'oracle'))
import torch
import bendy_tracer_amd as b
import bt_oracle_py as o

def compare(name, w, h, spp, n=0, output=0, seed=0x5EED):
    gs = b.Scene.load(f'scenes/{name}.json.gz'); cam = gs.find_by_tag('camera'); gs.set_camera_aspect(cam, w / h)
    buf = b.Buffer.new(w, h)
    tr = b.Tracer.with_config(b.Config(chunks_x=8, chunks_y=4, output=b.Output(output)))
    st = tr.render(gs, cam, b.RenderConfig.with_samples_subsample(spp, b.Subsample(n)), buf, seed=seed)
    torch.cuda.synchronize()
    stats = gs.last_stats()
    g = buf.numpy()[..., :3]
    os_ = o.Scene.load(f'scenes/{name}.json.gz'); oc = os_.find_by_tag('camera'); os_.set_camera_aspect(oc, w / h)
    res = {}
    for rec in (0, 1):
        cfg = o.default_config(samples=spp, subsample_n=n, output=output, recursive=rec)
        img, rc, seg = o.render(os_, oc, cfg, w, h, seed, nthreads=16)
        res[rec] = (img[..., :3], seg)
    tot = spp * max(1, n * n)
    d_it = np.abs(g - res[0][0]).max() / tot
    d_rec = np.abs(g - res[1][0]).max() / tot
    nbad = int((np.abs(g - res[0][0]).max(axis=-1) > 0).sum())
    print(f'{name:9s} {w}x{h}x{spp} n={n} out={output}: seg gpu={stats.segments} oracle={res[0][1]} '
          f'max|d| iter={d_it:.3e} rec={d_rec:.3e} pixels!=iter {nbad} kernel_ms={stats.kernel_ms:.3f}', flush=True)

if __name__ == '__main__':
    print(torch.cuda.get_device_name(0))
    compare('cornell', 64, 64, 1)
    compare('cornell', 256, 256, 1)
    compare('cornell2', 128, 128, 16)
    compare('scene', 192, 108, 16)
    compare('scene', 192, 108, 4, n=2)
    compare('volume', 192, 128, 8)
    compare('cloud', 192, 128, 8)
    for out in (1, 2, 3):
        compare('scene', 96, 54, 4, output=out)
        compare('volume', 96, 64, 4, output=out)
    # timing at full size
    for name, w, h, spp in [('scene', 1920, 1080, 64), ('cornell2', 512, 512, 16), ('volume', 1920, 1080, 64), ('cornell', 1920, 1080, 64)]:
        gs = b.Scene.load(f'scenes/{name}.json.gz'); cam = gs.find_by_tag('camera'); gs.set_camera_aspect(cam, w / h)
        tr = b.Tracer.with_config(b.Config(chunks_x=8, chunks_y=4))
        for it in range(3):
            buf = b.Buffer.new(w, h)
            torch.cuda.synchronize(); t = time.time()
            tr.render(gs, cam, b.RenderConfig.with_samples(spp), buf)
            torch.cuda.synchronize(); dt = time.time() - t
            st = gs.last_stats()
            print(f'{name} {w}x{h}x{spp}: wall {dt*1e3:.2f} ms kernel {st.kernel_ms:.2f} ms -> {w*h*spp/dt/1e6:.1f} Msamples/s; seg/sample {st.segments/st.samples:.3f}', flush=True)
