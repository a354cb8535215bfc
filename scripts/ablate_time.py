"""Times cornell 1024x768 (fast, wavefront) with all lights, and with light_count = 0 (extension rays only)."""
import sys, os
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "path-tracer_amd"))
import pyptgpu as pt
tag = os.path.basename(os.environ.get("PTGPU_LIB", "default"))
scn = sys.argv[1] if len(sys.argv) > 1 else "cornell"
sc = pt.LoadedScene(scn, 1024, 768); seeds = pt.seed_block(1024, 768)
bins = int(os.environ.get("BINS", "1"))
cull = {"auto": None, "tight": pt.CULL_TIGHT, "enclosed": pt.CULL_ENCLOSED, "ball": pt.CULL_BALL}[os.environ.get("CULL", "auto")]
for lights in (None, 0):
    g = pt.Gpu(0, 0, kernel=1, math=0, ray_bins=bins, shadow_cull=cull); g.set_scene(sc, light_count=lights); g.resize(1024, 768, seeds)
    n = 32
    g.render_timed(n)
    ms = min(g.render_timed(n) for _ in range(3))
    print(f"{tag} bins {bins} cull {os.environ.get('CULL', 'auto')} {scn} lights {'all' if lights is None else lights}: {ms / n:.3f} ms/pass  {1024 * 768 * n / ms / 1e3:.2f} Mpaths/s", flush=True)
    g.close()
