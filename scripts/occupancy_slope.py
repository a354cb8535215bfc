"""cornell 1024x768 (fast, wavefront): time per pass against waves per workgroup, i.e. against
resident waves per CU (two groups fit per CU from 5 waves per group up: 10, 12, 14, 16 waves/CU)."""
import sys, os
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "path-tracer_amd"))
import pyptgpu as pt
sc = pt.LoadedScene("cornell", 1024, 768); seeds = pt.seed_block(1024, 768)
for wpg in [int(a) for a in sys.argv[1:]] or (4, 5, 6, 7, 8, 9, 10, 12, 16):
    g = pt.Gpu(0, 0, kernel=1, math=0, waves_per_group=wpg); g.set_scene(sc); g.resize(1024, 768, seeds)
    n = 32
    g.render_timed(n)
    ms = min(g.render_timed(n) for _ in range(3))
    print(f"waves/group {wpg}: {ms / n:.3f} ms/pass  {1024 * 768 * n / ms / 1e3:.2f} Mpaths/s", flush=True)
    g.close()
