"""Small scenes on the wavefront kernel (path tracing): work queue on / off x passes per launch, 1024x768.  GPU box."""
import sys, os
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "path-tracer_amd"))
import pyptgpu as pt
W, H = 1024, 768
for name in sys.argv[1:] or ("materials", "simple-spheres", "stacked-spheres", "complex", "table-and-mirror", "obj-model"):
    sc = pt.LoadedScene(name, W, H); seeds = pt.seed_block(W, H); row = []
    for q, ppl, streams in ((2, 0, None), (1, 0, None), (0, 0, None), (0, 8, None), (0, 16, None), (0, 8, 1), (0, 32, 1)):
        g = pt.Gpu(0, 0, work_queue=q, passes_per_launch=ppl or None, streams=streams); g.set_scene(sc); g.resize(W, H, seeds)
        g.render_timed(64); ms = min(g.render_timed(64) / 64 for _ in range(4)); li = g.launch_info(); g.close()
        row.append(f"q{q}/{'1s' if streams else str(li['regions']) + 'r'}/{ppl or 'auto'}(u{li['unit_passes']}): {W * H / ms / 1e3:6.0f}")
    print(f"{name:18s} " + "  ".join(row), flush=True)
