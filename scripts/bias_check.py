"""Mean radiance of the library defaults (fast build) against the strict build at equal seeds, per scene, at a
sample count that resolves 1e-5 of the radiance: usage bias_check.py scene[:dl] ... (GPU box)."""
import sys, os
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "path-tracer_amd"))
import pyptgpu as pt
w, h, spp = 1024, 768, int(os.environ.get("SPP", "16"))
for arg in sys.argv[1:]:
    name, _, integ = arg.partition(":")
    sc = pt.LoadedScene(name, w, h); seeds = pt.seed_block(w, h)
    img = []
    for math in (pt.MATH_FAST, pt.MATH_STRICT):
        g = pt.Gpu(0, 1 if integ == "dl" else 0, math=math); g.set_scene(sc); g.resize(w, h, seeds); g.render(spp); g.finish()
        img.append(g.colors()); g.close()
    a, b = img
    ok = np.isfinite(a).all(1) & np.isfinite(b).all(1)
    d = a[ok].astype(np.float64) - b[ok]; m = b[ok].astype(np.float64).mean(0)
    per_px = d.mean(1)
    sigma = per_px.std() / np.sqrt(len(per_px)) / m.mean()
    print(f"{arg:22s} {w}x{h}x{spp}: bias rgb {(d.mean(0) / m).round(7).tolist()}  all {d.mean() / m.mean():+.2e}  (sigma of that mean {sigma:.1e})", flush=True)
