"""Strict build with PTGPU_CULL_BALL against the oracle's replay of the same sets: where do they differ?"""
import sys, os
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "path-tracer_amd")); sys.path.insert(0, os.path.join(REPO, "oracle"))
import pyptgpu as pt, oracle_binding as ob
w, h, spp = 96, 72, 1
sc = pt.LoadedScene("cornell", w, h)
keep = pt.shadow_candidates(sc, pt.CULL_BALL)
with ob.shadow_candidates(keep):
    cpu = ob.Render(sc, pt.seed_block(w, h), ob.SEEDS_KERNEL_LAYOUT, 0).pass_oracle(spp, threads=8)
for kernel in (pt.KERNEL_MEGA, pt.KERNEL_WAVEFRONT):
    for lds in (1, 0):
        g = pt.Gpu(0, 0, kernel=kernel, math=pt.MATH_STRICT, lds=lds, shadow_cull=pt.CULL_BALL)
        g.set_scene(sc); g.resize(w, h, pt.seed_block(w, h)); g.render(spp); g.finish()
        c = np.asarray(g.colors()).reshape(-1, 3); g.close()
        d = c.astype(np.float64) - cpu.colors
        bad = np.nonzero(np.nan_to_num(np.abs(d)).max(1) > 0)[0]
        print(f"kernel {kernel} lds {lds}: {len(bad)} of {w*h} pixels differ; gpu brighter {int((np.nan_to_num(d).sum(1) > 0).sum())} darker {int((np.nan_to_num(d).sum(1) < 0).sum())}; first {bad[:5]} d {d[bad[:3]]}", flush=True)
