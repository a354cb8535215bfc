"""BVH A/B: strict bit-exactness of BVH on vs off (and vs the oracle), then 1024x768 timing."""
import sys, os
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(REPO, "oracle")); sys.path.insert(0, os.path.join(REPO, "path-tracer_amd"))
import oracle_binding as ob, pyptgpu as pt
for name, w, h, spp in (("obj-model", 64, 48, 3), ("large-object", 40, 30, 1), ("complex", 64, 48, 2), ("cornell", 48, 36, 2),
                        ("table-and-mirror", 64, 48, 3), ("materials", 64, 48, 3), ("stacked-spheres", 64, 48, 3)):
    sc = pt.LoadedScene(name, w, h); seeds = pt.seed_block(w, h)
    cpu = ob.Render(sc, seeds, ob.SEEDS_KERNEL_LAYOUT, 0).pass_oracle(spp, threads=8)
    for k in (0, 1):
        for lds in (1, 0):
            g = pt.Gpu(0, 0, kernel=k, math=pt.MATH_STRICT, bvh=pt.BVH_ON, lds=lds); g.set_scene(sc); g.resize(w, h, seeds); g.render(spp)
            ok = g.colors().tobytes() == cpu.colors.tobytes() and (g.seeds() == cpu.seeds).all()
            info = g.scene_info()
            print(name, f"{w}x{h}x{spp} kernel {k} lds {lds} BVH on, strict bit-exact:", ok, "nodes", info["bvh_nodes"], "in tree", info["bvh_primitives"], "of", info["primitives"], flush=True); g.close()
w, h = 1024, 768
for scn in sys.argv[1:] or ["obj-model", "large-object", "complex", "table-and-mirror"]:
    sc = pt.LoadedScene(scn, w, h); seeds = pt.seed_block(w, h)
    for math in (0, 1):
        for kernel in (1, 0):
            for bvh in (0, 1):
                g = pt.Gpu(0, 0, kernel=kernel, math=math, bvh=bvh); g.set_scene(sc); g.resize(w, h, seeds)
                n = 8
                g.render_timed(n)
                ms = min(g.render_timed(n) for _ in range(2))
                print(f"{scn} {'wavefront' if kernel else 'mega'} {'strict' if math else 'fast'} bvh {bvh}: "
                      f"{ms / n:.2f} ms/pass  {w * h * n / ms / 1e3:.2f} Mpaths/s", flush=True)
                g.close()
