"""Counting build on the synthetic convex-cluster scene of tests/test_gpu_synthetic.py: are plane tests executed?"""
import sys, os
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(R, "path-tracer_amd")); sys.path.insert(0, os.path.join(R, "tests")); sys.path.insert(0, os.path.join(R, "oracle"))
import numpy as np, pyptgpu as pt
import test_gpu_synthetic as t
w, h = 128, 96
sc = t.make_scene(t.convex_scene(), w, h)
for bins in (1, 0):
    g = pt.Gpu(0, 0, lib=pt.variant_path("count"), math=pt.MATH_FAST, kernel=pt.KERNEL_WAVEFRONT, ray_bins=bins, shadow_cull=pt.CULL_TIGHT)
    g.set_scene(sc); g.resize(w, h, pt.seed_block(w, h)); g.work_counters(); g.render(4); g.finish()
    c = g.work_counters(); print(bins, g.scene_info(), g.launch_info(), {k: v for k, v in c.items() if v}); g.close()
