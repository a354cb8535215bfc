"""Sweep launch knobs of the wavefront kernel on one scene at 1024x768."""
import sys, os, itertools
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "path-tracer_amd"))
import pyptgpu as pt
scn = sys.argv[1] if len(sys.argv) > 1 else "cornell"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 16
sc = pt.LoadedScene(scn, 1024, 768); seeds = pt.seed_block(1024, 768)
for lds, waves, streams in itertools.product((1, 0), (4, 8, 12, 16), (4,)):
    g = pt.Gpu(0, 0, kernel=1, math=0, lds=lds, streams=streams, waves_per_group=waves); g.set_scene(sc); g.resize(1024, 768, seeds)
    g.render_timed(n)
    ms = min(g.render_timed(n) for _ in range(2))
    print(f"{scn} lds {lds} waves/group {waves} streams {streams}: {ms / n:.2f} ms/pass  {1024 * 768 * n / ms / 1e3:.2f} Mpaths/s", flush=True)
    g.close()
