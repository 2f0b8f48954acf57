import sys, time
sys.path.insert(0, "/root/repo")
import torch
import parallelraytracing_amd as prt
torch.cuda.set_device(0)
for name in ("RANDOM_BALLS_LARGE", "DEFAULT", "CORNELL"):
    scene = prt.Scene(name)
    W, H = 1920, 1080
    cam = prt.Camera(width=W, height=H)
    film = prt.Film(W, H)
    r = prt.HipWavefrontRenderer(device=0, max_depth=20)
    r.Init(film, scene, cam)
    r.set_samples_in_flight(16)
    r.render_async(16); r.synchronize(); r.reset_stats()
    t0 = time.perf_counter(); r.render_async(16); r.synchronize(); dt = time.perf_counter() - t0
    st = r.stats()
    print(f"{name}: {len(scene.primitives)} prims, 1080p x 16 spp, depth 20: {dt*1e3:.1f} ms = {dt/16*1e3:.2f} ms/spp, {st.rays_total/dt/1e6:.0f} Mrays/s, {st.rays_total/16/W/H:.2f} rays/pixel", flush=True)
    del r
