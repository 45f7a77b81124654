#!/bin/bash
cd $GRAFT_REPO_ROOT
./bendy_tracer_amd/bendy-tracer-hip --output full --width 640 --height 360 --samples 64 --subsample 2 --scene scenes/scene.json.gz --screenshot gpurun_out/lens_scene.png --lens 0.6,0.4,4.0,0.15,0.1,6.0,800 --quiet
./bendy_tracer_amd/bendy-tracer-hip --output full --width 640 --height 360 --samples 64 --subsample 2 --scene scenes/scene.json.gz --screenshot gpurun_out/flat_scene.png --quiet
python - <<'PY'
import time, torch
import bendy_tracer_amd as b
for lens in (None, dict(centre=(0.6,0.4,4.0), rs=0.15, step=0.1, radius=6.0, max_steps=800), dict(centre=(0.6,0.4,4.0), rs=0.15, step=0.25, radius=6.0, max_steps=800)):
    sc=b.Scene.load('scenes/scene.json.gz'); cam=sc.find_by_tag('camera'); sc.set_camera_aspect(cam,16/9)
    if lens: sc.set_lens(**lens)
    buf=b.Buffer.new(1920,1080)
    for i in range(3):
        b.Tracer.new().render(sc,cam,b.RenderConfig.with_samples(16),buf,sample_base=16*i)
        st=sc.last_stats()
    print('lens' if lens else 'flat', lens and lens['step'], f'kernel {st.kernel_ms:.2f} ms  {1920*1080*16/st.kernel_ms/1e3:.0f} Msamples/s  segments/sample {st.segments/st.samples:.3f}  RK4 steps/sample {st.lens_steps/st.samples:.1f}')
PY
