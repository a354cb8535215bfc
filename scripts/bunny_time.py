"""Throughput of data/scenes/obj-model-bunny.txt (5 050 triangles) at 1024x768: BVH on/off, both kernels."""
import sys, os
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "path-tracer_amd"))
import pyptgpu as pt
W, H = 1024, 768
name = sys.argv[1] if len(sys.argv) > 1 else "obj-model-bunny"
sc = pt.LoadedScene(name, W, H); seeds = pt.seed_block(W, H)
for integ in (0, 1):
    for math in (0, 1):
        for kernel in (1, 0):
            for bvh in (pt.BVH_AUTO, pt.BVH_OFF):
                g = pt.Gpu(0, integ, kernel=kernel, math=math, bvh=bvh); g.set_scene(sc); g.resize(W, H, seeds)
                info = g.scene_info()
                n = 4 if bvh == pt.BVH_OFF else 16
                g.render_timed(n)
                ms = min(g.render_timed(n) for _ in range(2))
                print(f"{name} {'pt' if integ == 0 else 'dl'} {'fast' if math == 0 else 'strict'} "
                      f"{'wavefront' if kernel == 1 else 'mega'} bvh={'auto' if bvh == pt.BVH_AUTO else 'off'} "
                      f"nodes={info['bvh_nodes']}: {W * H * n / ms / 1e3:.1f} Mpaths/s", flush=True)
                g.close()
