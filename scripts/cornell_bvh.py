import sys; sys.path.insert(0,'path-tracer_amd')
import pyptgpu as pt
sc = pt.LoadedScene("cornell", 1024, 768); seeds = pt.seed_block(1024, 768)
for bvh in (2, 1):
    g = pt.Gpu(0, 0, bvh=bvh); g.set_scene(sc); g.resize(1024, 768, seeds); g.render_timed(8)
    ms = min(g.render_timed(8) for _ in range(2)); print("cornell fast bvh", bvh, ms/8, "ms/pass"); g.close()
