"""Sweep launch knobs of the wavefront kernel on one scene at 1024x768.
usage: sweep.py <scene> <passes> <waves,waves,...> [lds list]   (PTGPU_LIB selects an A/B build)"""
import sys, os, itertools
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "path-tracer_amd"))
import pyptgpu as pt
scn = sys.argv[1] if len(sys.argv) > 1 else "cornell"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 16
waves_list = [int(x) for x in sys.argv[3].split(",")] if len(sys.argv) > 3 else [8]
lds_list = [int(x) for x in sys.argv[4].split(",")] if len(sys.argv) > 4 else [1]
tag = os.path.basename(os.environ.get("PTGPU_LIB", "default"))
sc = pt.LoadedScene(scn, 1024, 768); seeds = pt.seed_block(1024, 768)
for lds, waves in itertools.product(lds_list, waves_list):
    g = pt.Gpu(0, 0, kernel=1, math=0, lds=lds, streams=4, waves_per_group=waves); g.set_scene(sc); g.resize(1024, 768, seeds)
    g.render_timed(n)
    ms = min(g.render_timed(n) for _ in range(3))
    print(f"{tag} {scn} lds {lds} waves/group {waves}: {ms / n:.2f} ms/pass  {1024 * 768 * n / ms / 1e3:.2f} Mpaths/s", flush=True)
    g.close()
