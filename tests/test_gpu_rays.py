"""Ray-level checks through ptgpu_probe_rays (ray-probe build of the library, path-tracer_amd/variants/
libptgpu_probe.so): the intersection routines of every code instance on rays the test names, instead of on
whatever rays a render happens to trace.

* The tie rule (geomfunc.h:149: strict '<' in index order, so of two primitives at a bit-equal distance the lower
  index wins).  Round 1's verdict asked for a scene that shows what the fast build does instead; no scene file can
  force a tie on a jittered camera ray, a named ray can: a triangle in the plane z = 8 and a sphere tangent to that
  plane from behind, the ray along +z through the point of tangency - both distances are exactly 8.0f in both
  builds (every operation on the way is exact: powers of two).
* The hierarchy walks, per lane and wave-cooperative, against the linear scan on random rays: same distance bits,
  same surface, same occlusion answer.
"""
import numpy as np
import pytest

import pyptgpu as pt
from test_gpu_synthetic import make_scene, sphere, triangles

pytestmark = pytest.mark.gpu


def probe(sc, o, d, maxt, cooperative=False, **kw):
    g = pt.Gpu(0, 0, lib=pt.variant_path("probe"), **kw)
    g.set_scene(sc)
    g.resize(8, 8, pt.seed_block(8, 8))
    out = g.probe_rays(o, d, maxt, cooperative=cooperative)
    info = g.scene_info()
    g.close()
    return out + (info,)


def tie_scene(triangle_first):
    tri = triangles(np.array([[[-0.5, -0.5, 8], [1.5, -0.5, 8], [-0.5, 1.5, 8]]], dtype=np.float32), color=(.1, .9, .1))
    sph = sphere(4, (0, 0, 12), color=(.9, .1, .1))
    far = np.array([[[100 + i, 100, 100], [101 + i, 100, 100], [100 + i, 101, 100]] for i in range(8)], dtype=np.float32)
    rest = [triangles(far), sphere(1, (0, 50, 0), emission=(5, 5, 5))]
    return make_scene(([tri, sph] if triangle_first else [sph, tri]) + rest, 8, 8)


def test_bit_equal_distances_go_to_the_lower_index_in_the_strict_build_and_to_the_sphere_in_the_fast_one():
    o, d, far = [[0, 0, 0]], [[0, 0, 1]], [1e20]
    green, red = (.1, .9, .1), (.9, .1, .1)
    for triangle_first in (True, False):
        sc = tie_scene(triangle_first)
        # the reference's rule: index order decides - through the linear scan and through the hierarchy
        for kw in (dict(bvh=pt.BVH_OFF, shadow_cull=pt.CULL_OFF), dict(bvh=pt.BVH_ON), dict(bvh=pt.BVH_ON, cooperative=True)):
            t, rgb, occ, info = probe(sc, o, d, far, math=pt.MATH_STRICT, **kw)
            assert t[0] == np.float32(8.0) and occ[0], kw
            assert np.allclose(rgb[0], green if triangle_first else red), (triangle_first, kw, rgb)
        # the fast build traces in shadow order (spheres first): a sphere wins a tie against a triangle whatever
        # their indices - upstream's answer only when the sphere has the lower index
        for kw in (dict(bvh=pt.BVH_OFF), dict(bvh=pt.BVH_AUTO)):
            t, rgb, occ, info = probe(sc, o, d, far, math=pt.MATH_FAST, **kw)
            assert t[0] == np.float32(8.0) and occ[0], kw
            assert np.allclose(rgb[0], red), (triangle_first, kw, rgb)


def random_rays(n, seed):
    rng = np.random.default_rng(seed)
    o = rng.uniform((-45, 2, 5), (45, 75, 160), (n, 3)).astype(np.float32)
    d = rng.normal(size=(n, 3))
    d = (d / np.linalg.norm(d, axis=1)[:, None]).astype(np.float32)
    maxt = rng.uniform(1, 200, n).astype(np.float32)
    return o, d, maxt


@pytest.mark.parametrize("name", ["obj-model", "obj-model-bunny", "complex", "large-object", "cornell"])
def test_hierarchy_walks_answer_as_the_linear_scan_ray_by_ray(name):
    sc = pt.LoadedScene(name, 8, 8)
    o, d, maxt = random_rays(16384 + 37, 11)
    for math in (pt.MATH_STRICT, pt.MATH_FAST):
        lin = probe(sc, o, d, maxt, math=math, bvh=pt.BVH_OFF, shadow_cull=pt.CULL_OFF)
        assert (lin[0] > 0).mean() > 0.3                      # the rays do hit things (complex.txt is open to the sky)
        assert 0.01 < lin[2].mean() < 0.99                    # and the occlusion question is not one-sided
        for cooperative in (False, True):
            tree = probe(sc, o, d, maxt, math=math, bvh=pt.BVH_ON, cooperative=cooperative)
            assert tree[3]["bvh_nodes"] > 0
            assert (tree[0].view(np.uint32) == lin[0].view(np.uint32)).all(), (name, math, cooperative, "distance bits")
            assert (tree[1] == lin[1]).all(), (name, math, cooperative, "surface")
            assert (tree[2] == lin[2]).all(), (name, math, cooperative, "occlusion")


def test_face_plane_test_of_convex_clusters_answers_as_their_triangles_ray_by_ray():
    # The near batches of the fast build test a closed convex cluster by its face planes (radiance_device.inc,
    # segment_crosses_convex).  Ray by ray against the any-hit scan over the same clusters' triangles: random
    # segments through the room, and segments that START ON the meshes (a diffuse mesh starts its shadow rays on
    # itself), leaving outwards and diving inwards.  The two may differ only for a ray that grazes an edge within
    # rounding: none of these does.
    from test_gpu_synthetic import convex_scene
    rng = np.random.default_rng(5)
    for sc, boxes in ((make_scene(convex_scene(), 8, 8), [((-10, 0, 60), (10, 22, 80))]),
                      (pt.LoadedScene("cornell", 8, 8), [])):
        n = 20000
        o, d, maxt = random_rays(n, 3)
        if boxes:                                   # origins on the faces of the cube, any direction
            (x0, y0, z0), (x1, y1, z1) = boxes[0]
            p = rng.uniform((x0, y0, z0), (x1, y1, z1), (n // 2, 3))
            face = rng.integers(0, 6, n // 2)
            for a in range(3):
                p[face == 2 * a, a] = (x0, y0, z0)[a]
                p[face == 2 * a + 1, a] = (x1, y1, z1)[a]
            o[: n // 2] = p.astype(np.float32)
        g = pt.Gpu(0, 0, lib=pt.variant_path("probe"), math=pt.MATH_FAST)
        g.set_scene(sc)
        g.resize(8, 8, pt.seed_block(8, 8))
        assert g.scene_info()["convex_clusters"] >= 1
        _, _, bits = g.probe_rays(o, d, maxt, cooperative=2)
        g.close()
        planes, tris = (bits & 1) != 0, (bits & 2) != 0
        print(f"convex clusters: {tris.mean():.3f} of the segments cross one; plane test differs on {(planes != tris).sum()} of {n}")
        assert tris.mean() > 0.02 and (planes != tris).mean() <= 1e-4


def test_the_shipped_library_has_no_ray_probe():
    g = pt.Gpu(0, 0)
    g.set_scene(pt.LoadedScene("simple", 8, 8))
    g.resize(8, 8, pt.seed_block(8, 8))
    with pytest.raises(pt.GpuError, match="ray-probe build"):
        g.probe_rays([[0, 0, 0]], [[0, 0, 1]], [1.0])
    g.render(1)          # and the context is as usable as before
    g.finish()
    g.close()
