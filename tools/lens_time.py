"""Times the lens extension on C3's frame (developer tool)."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..'))
import bendy_tracer_amd as b
spp = int(os.environ.get('LENS_SPP', '16'))
for step in (0.1, 0.25):
    sc=b.Scene.load('scenes/scene.json.gz'); cam=sc.find_by_tag('camera'); sc.set_camera_aspect(cam,16/9); sc.tuning_from_env()
    sc.set_lens(centre=(0.6,0.4,4.0), rs=0.15, step=step, radius=6.0, max_steps=800)
    buf=b.Buffer.new(1920,1080); ks=[]
    for i in range(4):
        b.Tracer.new().render(sc,cam,b.RenderConfig.with_samples(spp),buf,sample_base=spp*i)
        st=sc.last_stats(); ks.append(st.kernel_ms)
    k=min(ks[1:])
    print(f'lens step {step}: kernel {k:.2f} ms  {1920*1080*spp/k/1e3:.0f} Msamples/s ({spp} spp, S={st.slices})  RK4 steps/sample {st.lens_steps/st.samples:.1f}', flush=True)
