"""Scene statistics as ptgpu_set_scene sees them, for every scene file of data/scenes."""
import sys, os
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "path-tracer_amd"))
import pyptgpu as pt
names = sys.argv[1:] or sorted(f[:-4] for f in os.listdir(os.path.join(REPO, "data", "scenes")))
for n in names:
    sc = pt.LoadedScene(n, 64, 48); g = pt.Gpu(0, 0); g.set_scene(sc); i = g.scene_info(); g.close()
    print(f"{n}: prims {i['primitives']} (sph {i['spheres']}, tri {i['triangles']}) lights {i['lights']} lds {i['lds_bytes']} "
          f"tests/ray safe {i['shadow_tests_safe']:.1f} tight {i['shadow_tests_tight']:.1f} enclosed {i['shadow_tests_enclosed']:.1f} "
          f"clusters {i['clusters']} bvh nodes {i['bvh_nodes']} prims {i['bvh_primitives']}")
