"""cornell (fast, wavefront, defaults) at 1024 x H: does throughput depend on how the tile count divides by the
wave slots of the chip (256 CUs x 24 waves = 6 144; 1024 x 768 is exactly 2.0 tiles per slot)?"""
import sys, os
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "path-tracer_amd"))
import pyptgpu as pt
W = 1024
for H in (384, 576, 768, 864, 960, 1152, 1536):
    sc = pt.LoadedScene("cornell", W, H); seeds = pt.seed_block(W, H)
    g = pt.Gpu(0, 0, kernel=1, math=0); g.set_scene(sc); g.resize(W, H, seeds)
    n = 32
    g.render_timed(n)
    ms = min(g.render_timed(n) for _ in range(3))
    tiles = (W // 8) * ((H + 7) // 8)
    print(f"1024x{H}: {tiles} tiles = {tiles / 6144:.2f} per wave slot: {W * H * n / ms / 1e3:.1f} Mpaths/s", flush=True)
    g.close()
