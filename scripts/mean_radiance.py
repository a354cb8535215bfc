"""Mean linear radiance of a scene after n passes (fast build, defaults) - to compare measurement variants."""
import sys, os
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "path-tracer_amd"))
import pyptgpu as pt
scn = sys.argv[1] if len(sys.argv) > 1 else "cornell"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 8
w, h = 1024, 768
sc = pt.LoadedScene(scn, w, h)
g = pt.Gpu(0, 0); g.set_scene(sc); g.resize(w, h, pt.seed_block(w, h))
g.render(n); g.finish()
c = np.asarray(g.colors()).reshape(-1, 3)
ok = np.isfinite(c).all(axis=1)
print(f"{os.path.basename(os.environ.get('PTGPU_LIB', 'default'))} {scn} x{n}: mean {c[ok].mean(dtype=np.float64):.6f}  nan pixels {int((~ok).sum())}", flush=True)
if len(sys.argv) > 3: np.save(sys.argv[3], c)
