"""Mega kernel (fast) on the one-light scenes at 1024x768 and cornell-large at 3840x2160."""
import sys, os
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "path-tracer_amd"))
import pyptgpu as pt
tag = os.path.basename(os.environ.get("PTGPU_LIB", "default"))
for scn, w, h in (("cornell-large", 3840, 2160), ("simple", 1024, 768), ("caustic", 1024, 768), ("materials", 1024, 768), ("obj-model", 1024, 768), ("cornell", 1024, 768)):
    sc = pt.LoadedScene(scn, w, h); seeds = pt.seed_block(w, h)
    for integ in (0, 1):
        g = pt.Gpu(0, integ, kernel=0, math=0, work_queue=int(os.environ.get('WQ', '2'))); g.set_scene(sc); g.resize(w, h, seeds)
        n = 64 if scn != "cornell" else 16
        g.render_timed(n)
        ms = min(g.render_timed(n) for _ in range(3))
        print(f"{tag} wq={os.environ.get('WQ', '2')} {scn} {w}x{h} {'pt' if integ == 0 else 'dl'} mega: {w * h * n / ms / 1e3:.1f} Mpaths/s", flush=True)
        g.close()
