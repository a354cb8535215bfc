"""One timed render of a scene at 1024x768 with a given library build (the program profilers are pointed at).
usage: render_once.py <lib.so> [scene[:dl]] [passes] [repeats]"""
import sys, os
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "path-tracer_amd"))
import pyptgpu as pt
lib = sys.argv[1]
name, _, integ = (sys.argv[2] if len(sys.argv) > 2 else "cornell").partition(":")
n = int(sys.argv[3]) if len(sys.argv) > 3 else 64
rep = int(sys.argv[4]) if len(sys.argv) > 4 else 2
W, H = 1024, 768
sc = pt.LoadedScene(name, W, H); seeds = pt.seed_block(W, H)
g = pt.Gpu(0, 1 if integ == "dl" else 0, lib=lib); g.set_scene(sc); g.resize(W, H, seeds)
for _ in range(rep):
    ms = g.render_timed(n)
    print(f"{name} {os.path.basename(lib)} {ms / n:.4f} ms/pass {W * H * n / ms / 1e3:.1f} Mpaths/s", flush=True)
print(g.launch_info() if hasattr(g, "launch_info") else "")
g.close()
