"""Executed work per path of counting builds of the library on one scene: usage count_compare.py scene lib.so [lib.so ...]"""
import sys, os
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "path-tracer_amd"))
import pyptgpu as pt
scene = sys.argv[1]
W, H, N = 256, 192, 4
sc = pt.LoadedScene(scene, W, H); seeds = pt.seed_block(W, H)
for lib in sys.argv[2:]:
    g = pt.Gpu(0, 0, lib=lib); g.set_scene(sc); g.resize(W, H, seeds)
    g.work_counters(reset=True)
    g.render(N); g.finish()
    c = g.work_counters()
    paths = max(c["paths"], 1)
    print(os.path.basename(lib), {k: round(v / paths, 3) for k, v in c.items() if v}, g.launch_info()["kernel"], flush=True)
    g.close()
