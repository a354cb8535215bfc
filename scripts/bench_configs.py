"""Throughput of every BASELINE.json configuration scene (and a few more) at 1024x768, both
integrators, fast and strict builds, default launch settings.  Prints a markdown table."""
import sys, os
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "path-tracer_amd"))
import pyptgpu as pt
W, H = 1024, 768
print("| scene | prims / lights | integrator | fast Mpaths/s | strict Mpaths/s | mega (fast) Mpaths/s |")
print("|---|---|---|---|---|---|")
for scn in ("cornell", "materials", "caustic", "obj-model", "simple", "cornell-large", "complex", "large-object", "table-and-mirror"):
    sc = pt.LoadedScene(scn, W, H); seeds = pt.seed_block(W, H)
    for integ in (0, 1):
        row = []
        for kernel, math in ((1, 0), (1, 1), (0, 0)):
            g = pt.Gpu(0, integ, kernel=kernel, math=math); g.set_scene(sc); g.resize(W, H, seeds)
            info = g.scene_info()
            n = 8 if info["primitives"] * (1 + info["lights"]) > 2000 else 64
            g.render_timed(n)
            ms = min(g.render_timed(n) for _ in range(2))
            row.append(W * H * n / ms / 1e3)
            g.close()
        print(f"| {scn} | {info['primitives']} / {info['lights']} | {'pt' if integ == 0 else 'dl'} | {row[0]:.1f} | {row[1]:.1f} | {row[2]:.1f} |", flush=True)
