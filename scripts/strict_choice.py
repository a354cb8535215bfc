"""Strict build (bit-exact with the reference): candidate sets (SAFE, the default) against the hierarchy for every ray
(PTGPU_OPT_BVH ON), per kernel, 1024x768.  Same bits either way (tests); which is faster?  GPU box."""
import sys, os
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "path-tracer_amd"))
import pyptgpu as pt
W, H = 1024, 768
for scn in sys.argv[1:] or ("cornell", "materials", "cornell-large", "obj-model", "simple"):
    sc = pt.LoadedScene(scn, W, H); seeds = pt.seed_block(W, H)
    row = []
    ref = None
    for kernel, bvh in ((pt.KERNEL_AUTO, pt.BVH_AUTO), (pt.KERNEL_WAVEFRONT, pt.BVH_ON), (pt.KERNEL_MEGA, pt.BVH_ON)):
        g = pt.Gpu(0, 0, math=pt.MATH_STRICT, kernel=kernel, bvh=bvh); g.set_scene(sc); g.resize(W, H, seeds)
        g.render_timed(2); ms = min(g.render_timed(4) / 4 for _ in range(2))
        g.finish(); img = g.colors().tobytes()
        if ref is None: ref = img
        row.append(f"{W * H / ms / 1e3:8.1f} ({'same' if img == ref else 'DIFFERENT'} bits)"); g.close()
    print(f"{scn:16s} strict: default {row[0]}  wavefront+BVH {row[1]}  mega+BVH {row[2]}", flush=True)
