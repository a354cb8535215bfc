"""Every scene file of the reference, both integrators, library defaults (fast build, AUTO everything) against the
strict build (= the reference's CPU path bit for bit, tests/test_gpu_parity.py) at equal seeds: what was launched,
throughput at 1024x768, and the image difference at 512x384 x 8 spp.  GPU box; writes a markdown table."""
import sys, os, glob
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "path-tracer_amd"))
import pyptgpu as pt
out = open(sys.argv[1] if len(sys.argv) > 1 else os.path.join(REPO, "gpurun_out", "all_scenes.md"), "w")
def emit(s):
    print(s, flush=True); out.write(s + "\n"); out.flush()
emit("| scene | integrator | kernel / instance / sets | Mpaths/s 1024x768 | mean bias vs strict | rel. RMSE | 8x8-block rel. err p99 | seeds equal | NaN px fast / strict |")
emit("|---|---|---|---|---|---|---|---|---|")
INST = {0: "plain", 1: "bvh", 2: "hybrid", 3: "ball"}
CULL = {v: k for k, v in dict(off=pt.CULL_OFF, safe=pt.CULL_SAFE, tight=pt.CULL_TIGHT, enclosed=pt.CULL_ENCLOSED, ball=pt.CULL_BALL).items()}
for path in sorted(glob.glob(os.path.join(REPO, "data", "scenes", "*.txt"))):
    name = os.path.basename(path)[:-4]
    for integ in (0, 1):
        w, h, spp = 512, 384, 8
        sc = pt.LoadedScene(name, w, h); seeds = pt.seed_block(w, h)
        img = []
        for math in (pt.MATH_FAST, pt.MATH_STRICT):
            g = pt.Gpu(0, integ, math=math); g.set_scene(sc); g.resize(w, h, seeds); g.render(spp); g.finish()
            img.append((g.colors(), g.seeds())); g.close()
        (a, sa), (b, sb) = img
        ok = np.isfinite(a).all(1) & np.isfinite(b).all(1)
        d = a[ok].astype(np.float64) - b[ok]; mean = max(float(b[ok].mean()), 1e-9)
        blk = 8
        A = np.where(ok[:, None], a, 0).reshape(h, w, 3).astype(np.float64); B = np.where(ok[:, None], b, 0).reshape(h, w, 3).astype(np.float64)
        Ab = A.reshape(h // blk, blk, w // blk, blk, 3).mean((1, 3, 4)); Bb = B.reshape(h // blk, blk, w // blk, blk, 3).mean((1, 3, 4))
        p99 = float(np.quantile(np.abs(Ab - Bb) / np.maximum(Bb, 1e-3), .99))
        W, H = 1024, 768
        sc2 = pt.LoadedScene(name, W, H); g = pt.Gpu(0, integ); g.set_scene(sc2); g.resize(W, H, pt.seed_block(W, H))
        info = g.scene_info(); n = 16 if info["primitives"] * (1 + info["lights"]) > 2000 else 64
        g.render_timed(n); ms = min(g.render_timed(n) for _ in range(3)); li = g.launch_info(); g.close()
        emit(f"| {name} | {'dl' if integ else 'pt'} | {'wavefront' if li['kernel'] == pt.KERNEL_WAVEFRONT else 'mega'} / {INST[li['instance']]} / {CULL[li['shadow_cull']]} | "
             f"{W * H * n / ms / 1e3:.0f} | {d.mean() / mean:+.1e} | {100 * np.sqrt((d ** 2).mean()) / mean:.2f} % | {100 * p99:.2f} % | "
             f"{100 * (sa.reshape(-1, 2) == sb.reshape(-1, 2)).all(1).mean():.3f} % | {int((~np.isfinite(a).all(1)).sum())} / {int((~np.isfinite(b).all(1)).sum())} |")
