"""Developer tool: launches of T = 1 ... 64 rays per pixel per call on a frame (BT_FRAME, default 1920x1080) under several
pixel-block sizes -- the data behind bt_api.cpp's choice of block size (the reference's interactive pattern,
main.rs:245-254, is T = 1 sample x Subpixel(2) = 4).  Prints ms per pipelined call (wall) and the kernel's HIP-event time."""
import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..'))
import torch
import bendy_tracer_amd as b
w, h = (int(x) for x in os.environ.get('BT_FRAME', '1920x1080').split('x'))
names = os.environ.get('BT_ONLY', 'scene,cornell2,volume').split(',')
shapes = [(1, 1), (2, 1), (1, 2), (8, 1), (16, 1), (32, 1), (64, 1)]
if os.environ.get('BT_T'):
    shapes = [(int(t), 1) if t != '4' else (1, 2) for t in os.environ['BT_T'].split(',')]
modes = os.environ.get('BT_MODES', 'auto,s1,s2,s4,s8,p32').split(',')     # u = one block per workgroup; sN = the same with blocks of 256/N pixels; p / pN = packed launch (with such blocks); c / cN = packed with the compacting drain
for name in names:
    sc = b.Scene.load(f'scenes/{name}.json.gz'); cam = sc.find_by_tag('camera'); sc.set_camera_aspect(cam, w / h)
    tr = b.Tracer.with_config(b.Config(chunks_x=8, chunks_y=4))
    for samples, sub in shapes:
        row = []
        for mode in modes:
            sc.set_tuning()
            if mode == 'u':
                sc.set_tuning(packed=0)
            elif mode[0] == 's':
                sc.set_tuning(slices=int(mode[1:]), packed=0)
            elif mode[0] == 'p':
                sc.set_tuning(packed=1, slices=int(mode[1:] or 0))
            elif mode[0] == 'v':                          # automatic shape, phase vote pinned to N iterations
                sc.set_tuning(phase_vote=int(mode[1:]))
            elif mode[0] == 'c':
                sc.set_tuning(packed=2, slices=int(mode[1:] or 0))
            buf = b.Buffer.new(w, h)
            rc = b.RenderConfig.with_samples_subsample(samples, b.Subsample(sub))
            try:
                ks = []
                for i in range(4):
                    tr.render(sc, cam, rc, buf)
                    ks.append(sc.last_stats().kernel_ms)
                torch.cuda.synchronize()
                t = time.perf_counter()
                n = 30
                for i in range(n):
                    tr.render(sc, cam, rc, buf)
                torch.cuda.synchronize()
                dt = (time.perf_counter() - t) / n
                st = sc.last_stats()
                row.append(f'{mode}:{dt*1e3:.3f}/k{min(ks[1:]):.3f}(S{st.slices})')
            except Exception as e:
                row.append(f'{mode}:err {str(e)[:40]}')
        print(f'{name:9s} {w}x{h} T={samples*sub*sub:3d} ms per call  ' + '  '.join(row), flush=True)
