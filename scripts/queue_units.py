"""Work queue with units of several passes (PTGPU_UNIT_PASSES), forced on, against the defaults."""
import sys, os
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "path-tracer_amd"))
import pyptgpu as pt
W, H = 1024, 768
q = int(sys.argv[1])
for scn in sys.argv[2:]:
    sc = pt.LoadedScene(scn, W, H); seeds = pt.seed_block(W, H)
    g = pt.Gpu(0, 0, kernel=1, work_queue=q); g.set_scene(sc); g.resize(W, H, seeds)
    n = 64
    g.render_timed(n)
    ms = min(g.render_timed(n) for _ in range(3))
    print(f"{scn} queue={q} unit_passes={os.environ.get('PTGPU_UNIT_PASSES', '1')}: {W * H * n / ms / 1e3:.0f} Mpaths/s", flush=True)
    g.close()
