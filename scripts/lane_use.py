"""Lane utilisation of the hierarchy walks and primitive tests (counting build), per scene.  GPU box."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "path-tracer_amd"))
import pyptgpu as pt
for scene in sys.argv[1:] or ["obj-model", "obj-model-bunny", "large-object", "complex", "cornell"]:
    w, h = 256, 192
    g = pt.Gpu(0, 0, lib=pt.variant_path("count"))
    g.set_scene(pt.LoadedScene(scene, w, h)); g.resize(w, h, pt.seed_block(w, h))
    g.work_counters(); g.render(2); g.finish(); c = g.work_counters(); li = g.launch_info(); g.close()
    p = c["paths"]; tests = c["sphere_tests"] + c["triangle_tests_mt"] + c["triangle_tests_planes"]
    print(f"{scene:18s} kernel {li['kernel']} inst {li['instance']}: per path nodes {c['bvh_nodes']/p:8.1f} (lane use {c['bvh_nodes']/max(c['node_lane_slots'],1):.3f})  "
          f"prim tests {tests/p:8.1f} (lane use {tests/max(c['prim_lane_slots'],1):.3f})  ext {c['extension_rays']/p:.2f} shadow {c['shadow_rays']/p:.2f}", flush=True)
