"""Scenes that run without the work queue (mega kernel; direct lighting): passes inside one launch
(PTGPU_OPT_PASSES_PER_LAUNCH; the image state stays in registers across them) and region streams
(PTGPU_OPT_STREAMS) against throughput at 1024x768, 64 passes per render call.  GPU box."""
import sys, os
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "path-tracer_amd"))
import pyptgpu as pt
W, H = 1024, 768
for arg in sys.argv[1:] or ("caustic", "caustic:dl", "materials:dl", "simple", "cornell-large", "skybox", "cornell:dl", "obj-model:dl",
                            "complex:dl", "table-and-mirror:dl", "stacked-spheres:dl", "simple-spheres:dl", "cornell-large:dl"):
    name, _, integ = arg.partition(":")
    sc = pt.LoadedScene(name, W, H); seeds = pt.seed_block(W, H); row = []
    for streams, ppl in ((None, 0), (None, 4), (None, 8), (None, 16), (1, 8), (1, 16), (1, 32), (1, 64)):
        g = pt.Gpu(0, 1 if integ == "dl" else 0, passes_per_launch=ppl or None, streams=streams); g.set_scene(sc); g.resize(W, H, seeds)
        g.render_timed(64); ms = min(g.render_timed(64) / 64 for _ in range(4)); li = g.launch_info(); g.close()
        row.append(f"{'1s' if streams else str(li['regions']) + 'r'}/{ppl or 'auto'}: {W * H / ms / 1e3:6.0f}")
    print(f"{arg:20s} " + "  ".join(row), flush=True)
