"""Shadow-cull A/B: strict bit-exactness of SAFE/TIGHT against OFF and the oracle, then cornell 1024x768 timing."""
import sys, os
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(REPO, "oracle")); sys.path.insert(0, os.path.join(REPO, "path-tracer_amd"))
import oracle_binding as ob, pyptgpu as pt
for name, w, h, spp in (("cornell", 40, 24, 2), ("materials", 64, 48, 3), ("cornell", 96, 72, 4)):
    sc = pt.LoadedScene(name, w, h); seeds = pt.seed_block(w, h)
    cpu = ob.Render(sc, seeds, ob.SEEDS_KERNEL_LAYOUT, 0).pass_oracle(spp, threads=8)
    for k in (0, 1):
        for cull in (pt.CULL_OFF, pt.CULL_SAFE, pt.CULL_TIGHT):
            g = pt.Gpu(0, 0, kernel=k, math=pt.MATH_STRICT, shadow_cull=cull); g.set_scene(sc); g.resize(w, h, seeds); g.render(spp)
            ok = g.colors().tobytes() == cpu.colors.tobytes() and (g.seeds() == cpu.seeds).all()
            print(name, f"{w}x{h}x{spp} kernel", k, "cull", cull, "strict bit-exact:", ok, g.scene_info(), flush=True); g.close()
w, h = 1024, 768
for scn in sys.argv[1:] or ["cornell"]:
    sc = pt.LoadedScene(scn, w, h); seeds = pt.seed_block(w, h)
    ref = None
    for math in (1, 0):
        for kernel in (1, 0):
            for cull in (0, 1, 2):
                g = pt.Gpu(0, 0, kernel=kernel, math=math, shadow_cull=cull); g.set_scene(sc); g.resize(w, h, seeds)
                n = 8
                g.render_timed(n)
                col = g.colors().copy()
                ms = min(g.render_timed(n) for _ in range(2))
                same = ""
                if math == 1:
                    if cull == 0 and kernel == 1: ref = col
                    same = f" identical-to-off: {col.tobytes() == ref.tobytes()} (differing px {(col != ref).any(1).sum()})"
                print(f"{scn} {'wavefront' if kernel else 'mega'} {'strict' if math else 'fast'} cull {cull}: "
                      f"{ms / n:.2f} ms/pass  {w * h * n / ms / 1e3:.2f} Mpaths/s{same}", flush=True)
                g.close()
