"""First GPU contact: mega kernel, strict and fast, against the CPU oracle on small images."""
import sys, time, os
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(REPO, "oracle")); sys.path.insert(0, os.path.join(REPO, "path-tracer_amd"))
import oracle_binding as ob, pyptgpu as pt

def run(name, w, h, spp, integ, math, kernel=pt.KERNEL_MEGA, lds=1):
    sc = pt.LoadedScene(name, w, h); seeds = pt.seed_block(w, h)
    cpu = ob.Render(sc, seeds, ob.SEEDS_KERNEL_LAYOUT, integ).pass_oracle(spp, threads=8)
    g = pt.Gpu(0, integ, kernel=kernel, math=math, lds=lds); g.set_scene(sc); g.resize(w, h, seeds)
    t0 = time.time(); g.render(spp); g.finish(); dt = time.time() - t0
    col, pix, sd = g.colors(), g.pixels(), g.seeds()
    seed_eq = (sd.reshape(-1, 2) == cpu.seeds.reshape(-1, 2)).all(1)
    both = ~(np.isnan(col).any(1) | np.isnan(cpu.colors).any(1))
    d = np.abs(col[both] - cpu.colors[both])
    bit = (col.view(np.uint32) == cpu.colors.view(np.uint32)).all(1).mean()
    print(f"{name:14s} {w}x{h}x{spp} integ={integ} math={'strict' if math else 'fast'} lds={lds}: seed-agree {seed_eq.mean()*100:.3f}% bit-equal px {bit*100:.3f}% "
          f"pixels-equal {(pix==cpu.pixels).mean()*100:.2f}% maxabs {d.max():.3g} rmse {np.sqrt((d**2).mean()):.3g} nan gpu/cpu {int(np.isnan(col).any(1).sum())}/{int(np.isnan(cpu.colors).any(1).sum())} {dt*1e3:.1f} ms", flush=True)
    g.close()

K = pt.KERNEL_WAVEFRONT
for math in (pt.MATH_STRICT, pt.MATH_FAST):
    for lds in (1, 0):
        run("cornell-large", 64, 48, 4, 0, math, K, lds=lds)
    run("simple", 64, 48, 4, 0, math, K)
    run("caustic", 64, 48, 8, 0, math, K)
    run("materials", 64, 48, 4, 0, math, K)
    run("materials", 64, 48, 4, 1, math, K)
    run("obj-model", 32, 24, 2, 0, math, K)
    run("cornell", 32, 24, 2, 0, math, K)
    run("cornell", 37, 21, 2, 0, math, K, lds=0)
    run("complex", 32, 24, 2, 0, math, K)
sc = pt.LoadedScene("cornell", 1024, 768); seeds = pt.seed_block(1024, 768)
for kernel in (pt.KERNEL_MEGA, pt.KERNEL_WAVEFRONT):
  for math in (pt.MATH_STRICT, pt.MATH_FAST):
    for lds in (1, 0):
        g = pt.Gpu(0, 0, kernel=kernel, math=math, lds=lds); g.set_scene(sc); g.resize(1024, 768, seeds)
        g.render_timed(1)
        ms = g.render_timed(4)
        print("cornell 1024x768 kernel", kernel, "math", math, "lds", lds, "ms/pass", ms / 4, "Mpaths/s", 1024 * 768 * 4 / ms / 1e3, flush=True)
        g.close()
