"""Mega against wavefront kernel per scene at 1024x768 (fast build, everything else default): the basis of
PTGPU_KERNEL_AUTO's rule for hierarchy scenes (ptgpu_api.hip).  GPU box."""
import sys, os
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "path-tracer_amd"))
import pyptgpu as pt
W, H = 1024, 768
INTEG = int(os.environ.get("INTEGRATOR", "0"))
for scn in sys.argv[1:] or ("complex", "table-and-mirror", "large-object", "obj-model", "obj-model-bunny", "stacked-spheres"):
    sc = pt.LoadedScene(scn, W, H); seeds = pt.seed_block(W, H)
    gs = []
    for kernel in (pt.KERNEL_MEGA, pt.KERNEL_WAVEFRONT, pt.KERNEL_AUTO):
        g = pt.Gpu(0, INTEG, kernel=kernel); g.set_scene(sc); g.resize(W, H, seeds)
        g.render_timed(16)
        gs.append(g)
    best = [1e30] * 3
    for _ in range(4):
        for i, g in enumerate(gs):
            best[i] = min(best[i], g.render_timed(32) / 32)
    info = gs[0].scene_info()
    print(f"{scn:18s} bvh prims {info.get('bvh_primitives', 0):6d}  mega {W*H/best[0]/1e3:8.1f}  wavefront {W*H/best[1]/1e3:8.1f}  auto {W*H/best[2]/1e3:8.1f} Mpaths/s", flush=True)
    for g in gs: g.close()
