"""cornell 1024x768 (fast, wavefront): time per pass against passes per launch and region streams."""
import sys, os
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "path-tracer_amd"))
import pyptgpu as pt
sc = pt.LoadedScene("cornell", 1024, 768); seeds = pt.seed_block(1024, 768)
for streams in (4, 2, 1):
    for ppl in (1, 2, 4, 8):
        g = pt.Gpu(0, 0, kernel=1, math=0, streams=streams, passes_per_launch=ppl); g.set_scene(sc); g.resize(1024, 768, seeds)
        n = 32
        g.render_timed(n)
        ms = min(g.render_timed(n) for _ in range(3))
        print(f"streams {streams} passes/launch {ppl}: {ms / n:.3f} ms/pass  {1024 * 768 * n / ms / 1e3:.2f} Mpaths/s", flush=True)
        g.close()
