import sys, os
sys.path.insert(0, "path-tracer_amd")
import pyptgpu as pt
W, H = 1024, 768
sc = pt.LoadedScene("cornell", W, H); seeds = pt.seed_block(W, H)
for cull in (pt.CULL_ENCLOSED, pt.CULL_BALL):
    for lib in (pt.variant_path("base"), None):
        g = pt.Gpu(0, 0, shadow_cull=cull, lib=lib); g.set_scene(sc); g.resize(W, H, seeds); g.render(2); g.finish()
        print(cull, lib, g.launch_info(), flush=True); g.close()
