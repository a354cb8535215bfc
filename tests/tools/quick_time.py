"""Quick A/B: parity spot check (strict bit-exact) + cornell 1024x768 timing of the kernel variants."""
import sys, os
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(REPO, "oracle")); sys.path.insert(0, os.path.join(REPO, "path-tracer_amd"))
import oracle_binding as ob, pyptgpu as pt
for name, w, h, spp in (("cornell", 40, 24, 2), ("materials", 64, 48, 3), ("obj-model", 32, 24, 2)):
    sc = pt.LoadedScene(name, w, h); seeds = pt.seed_block(w, h)
    cpu = ob.Render(sc, seeds, ob.SEEDS_KERNEL_LAYOUT, 0).pass_oracle(spp, threads=8)
    for k in (0, 1):
        g = pt.Gpu(0, 0, kernel=k, math=pt.MATH_STRICT); g.set_scene(sc); g.resize(w, h, seeds); g.render(spp)
        ok = g.colors().tobytes() == cpu.colors.tobytes() and (g.seeds() == cpu.seeds).all()
        print(name, "kernel", k, "strict bit-exact:", ok, flush=True); g.close()
scenes = sys.argv[1:] or ["cornell"]
for scn in scenes:
    sc = pt.LoadedScene(scn, 1024, 768); seeds = pt.seed_block(1024, 768)
    for kernel, math, streams, ppl in ((0, 0, 4, 1), (1, 1, 4, 1), (1, 0, 1, 8), (1, 0, 4, 1)):
            g = pt.Gpu(0, 0, kernel=kernel, math=math, streams=streams, passes_per_launch=ppl); g.set_scene(sc); g.resize(1024, 768, seeds)
            n = 16 if scn in ("cornell", "obj-model", "large-object") else 64
            g.render_timed(n)
            ms = min(g.render_timed(n) for _ in range(2))
            print(f"{scn} 1024x768 kernel {'wavefront' if kernel else 'mega'} math {'strict' if math else 'fast'} streams {streams} passes/launch {ppl}: "
                  f"{ms / n:.2f} ms/pass  {1024 * 768 * n / ms / 1e3:.2f} Mpaths/s", flush=True)
            g.close()
