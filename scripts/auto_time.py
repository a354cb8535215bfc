"""Default settings (kernel AUTO, math fast) per scene at 1024x768: what a caller who sets nothing gets."""
import sys, os
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "path-tracer_amd"))
import pyptgpu as pt
W, H = 1024, 768
for scn in sys.argv[1:] or ("cornell", "materials", "caustic", "obj-model", "obj-model-bunny", "simple", "cornell-large", "complex",
                            "large-object", "table-and-mirror"):
    sc = pt.LoadedScene(scn, W, H); seeds = pt.seed_block(W, H)
    row = []
    for integ in (0, 1):
        g = pt.Gpu(0, integ); g.set_scene(sc); g.resize(W, H, seeds)
        info = g.scene_info()
        n = 8 if info["primitives"] * (1 + info["lights"]) > 2000 else 64
        g.render_timed(n)
        ms = min(g.render_timed(n) for _ in range(3))
        row.append(W * H * n / ms / 1e3)
        g.close()
    print(f"{scn}: pt {row[0]:.0f}  dl {row[1]:.0f} Mpaths/s (defaults)", flush=True)
