"""A/B timing on one GPU box: each library named on the command line (default: the base build and the working
tree's) renders the same cases, interleaved over several rounds; prints the best ms per pass of each.
usage: ab_time.py [--cases cornell,materials,...] [--cull ball] [lib.so ...]"""
import sys, os
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "path-tracer_amd"))
import pyptgpu as pt
args = sys.argv[1:]
cases = ["cornell"]
kw = {}
waves = None
while args and args[0] in ("--cases", "--cull", "--waves"):
    if args[0] == "--cases": cases = args[1].split(",")
    elif args[0] == "--waves": waves = [int(x) for x in args[1].split(",")]   # waves per group, one per library
    else: kw["shadow_cull"] = {"off": pt.CULL_OFF, "safe": pt.CULL_SAFE, "tight": pt.CULL_TIGHT, "enclosed": pt.CULL_ENCLOSED, "ball": pt.CULL_BALL}[args[1]]
    args = args[2:]
libs = args or [pt.variant_path("base"), os.path.join(REPO, "path-tracer_amd", "libptgpu.so")]
W, H = 1024, 768
for scn in cases:
    name, _, integ = scn.partition(":")
    sc = pt.LoadedScene(name, W, H); seeds = pt.seed_block(W, H)
    gs = []
    for li, lib in enumerate(libs):
        if waves: kw["waves_per_group"] = waves[li] or None
        g = pt.Gpu(0, 1 if integ == "dl" else 0, lib=lib, **kw); g.set_scene(sc); g.resize(W, H, seeds)
        info = g.scene_info()
        n = 32 if info["primitives"] * (1 + info["lights"]) > 2000 else 128
        g.render_timed(n)
        gs.append(g)
    best = [1e30] * len(libs)
    for _ in range(4):
        for i, g in enumerate(gs):
            best[i] = min(best[i], g.render_timed(n) / n)
    for i, lib in enumerate(libs):
        print(f"{scn:22s} {os.path.basename(lib):28s} {best[i]:9.4f} ms/pass {W * H / best[i] / 1e3:10.1f} Mpaths/s  ({best[i] / best[0]:.4f} x first)", flush=True)
    for g in gs:
        g.close()
