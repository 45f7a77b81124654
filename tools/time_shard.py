"""Per-rank kernel time under bench.py's weak scaling, measured on one GPU (developer tool).
usage: time_shard.py [scene] [bt_tuning.slices values, comma separated; "auto" = unset]"""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..'))
import torch
import bendy_tracer_amd as b
w, h = 1920, 1080
name = sys.argv[1] if len(sys.argv) > 1 else 'scene'
modes = sys.argv[2].split(',') if len(sys.argv) > 2 else ['auto', '1']
sc = b.Scene.load(f'scenes/{name}.json.gz'); cam = sc.find_by_tag('camera'); sc.set_camera_aspect(cam, w / h)
tr = b.Tracer.with_config(b.Config(chunks_x=8, chunks_y=4))
for mode in modes:
    sc.set_tuning(slices=0 if mode == 'auto' else int(mode))
    for world in (1, 2, 4, 8):
        spp = 64 * world
        for rank in sorted({0, world - 1}):
            shard = b.new_shard(w, h, world); ks = []
            for it in range(4):
                tr.render_shard(sc, cam, b.RenderConfig.with_samples(spp), shard, w, h, rank, world, sample_base=it * spp)
                torch.cuda.synchronize()
                ks.append(sc.last_stats().kernel_ms)
            st = sc.last_stats()
            print(f'{name} slices={mode:>4} ({st.slices:2d}) world {world} rank {rank}: {spp} spp, kernel {min(ks[1:]):7.3f} ms, '
                  f'{st.samples / min(ks[1:]) / 1e6:8.1f} Gsamples/s, seg/sample {st.segments/st.samples:.3f}', flush=True)
