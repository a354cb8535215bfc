"""Sweep streams x passes-per-launch x waves-per-group of the wavefront kernel on one scene at 1024x768."""
import sys, os, itertools
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "path-tracer_amd"))
import pyptgpu as pt
scn = sys.argv[1] if len(sys.argv) > 1 else "cornell"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 32
sc = pt.LoadedScene(scn, 1024, 768); seeds = pt.seed_block(1024, 768)
for streams, ppl, waves in itertools.product((1, 2, 4), (1, 2, 4, 8), (0, 4, 8, 16)):
    g = pt.Gpu(0, 0, kernel=1, math=0, streams=streams, passes_per_launch=ppl, waves_per_group=waves); g.set_scene(sc); g.resize(1024, 768, seeds)
    g.render_timed(n)
    ms = min(g.render_timed(n) for _ in range(3))
    print(f"{scn} streams {streams} passes/launch {ppl} waves/group {waves}: {ms / n:.3f} ms/pass  {1024 * 768 * n / ms / 1e3:.2f} Mpaths/s", flush=True)
    g.close()
