"""A/B of one library build over a set of (scene, integrator, kernel, math) cases at 1024x768: prints
Mpaths/s per case.  PTGPU_LIB selects the build; run once per build on the same box."""
import sys, os
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "path-tracer_amd"))
import pyptgpu as pt
W, H = 1024, 768
tag = os.path.basename(os.environ.get("PTGPU_LIB", "default"))
cases = [("cornell", 0, 1, 1), ("cornell", 0, 0, 0), ("materials", 0, 1, 1), ("materials", 0, 1, 0), ("materials", 1, 1, 0),
         ("materials", 0, 0, 0), ("caustic", 0, 1, 1), ("caustic", 0, 1, 0), ("caustic", 0, 0, 0), ("simple", 0, 0, 0),
         ("cornell-large", 0, 0, 0), ("obj-model", 0, 1, 0)]
for scn, integ, kernel, math in cases:
    sc = pt.LoadedScene(scn, W, H); seeds = pt.seed_block(W, H)
    g = pt.Gpu(0, integ, kernel=kernel, math=math); g.set_scene(sc); g.resize(W, H, seeds)
    info = g.scene_info()
    n = 8 if info["primitives"] * (1 + info["lights"]) > 2000 else 64
    g.render_timed(n)
    ms = min(g.render_timed(n) for _ in range(3))
    print(f"{tag:22s} {scn:14s} {'pt' if integ == 0 else 'dl'} {'wavefront' if kernel else 'mega':9s} {'strict' if math else 'fast':6s} {W * H * n / ms / 1e3:10.1f}", flush=True)
    g.close()
