"""Synthetic scenes that push the shadow-ray tables and the hierarchies past what the reference's scene
files reach: several mask words per light, more than 8 mesh clusters, duplicated geometry (exact
distance ties), degenerate and non-finite primitives.  Bars as in test_gpu_parity.py: the strict
build is bit-exact against the CPU oracle in every mode; the fast build's modes agree with each other.
"""
import numpy as np
import pytest

import oracle_binding as ob
import pyptgpu as pt
from conftest import same_bits

pytestmark = pytest.mark.gpu


def icosphere(center, radius, subdiv):
    t = (1 + 5 ** 0.5) / 2
    v = [(-1, t, 0), (1, t, 0), (-1, -t, 0), (1, -t, 0), (0, -1, t), (0, 1, t), (0, -1, -t), (0, 1, -t),
         (t, 0, -1), (t, 0, 1), (-t, 0, -1), (-t, 0, 1)]
    f = [(0, 11, 5), (0, 5, 1), (0, 1, 7), (0, 7, 10), (0, 10, 11), (1, 5, 9), (5, 11, 4), (11, 10, 2), (10, 7, 6),
         (7, 1, 8), (3, 9, 4), (3, 4, 2), (3, 2, 6), (3, 6, 8), (3, 8, 9), (4, 9, 5), (2, 4, 11), (6, 2, 10),
         (8, 6, 7), (9, 8, 1)]
    v = [np.array(p, dtype=np.float64) / np.linalg.norm(p) for p in v]
    for _ in range(subdiv):
        nf, cache = [], {}

        def mid(a, b):
            key = (min(a, b), max(a, b))
            if key not in cache:
                m = v[a] + v[b]
                v.append(m / np.linalg.norm(m))
                cache[key] = len(v) - 1
            return cache[key]
        for a, b, c in f:
            ab, bc, ca = mid(a, b), mid(b, c), mid(c, a)
            nf += [(a, ab, ca), (b, bc, ab), (c, ca, bc), (ab, bc, ca)]
        f = nf
    v = np.array(v) * radius + np.array(center)
    return np.array([[v[a], v[b], v[c]] for a, b, c in f], dtype=np.float32)


def triangles(tris, emission=(0, 0, 0), color=(.7, .7, .7), material=0):
    o = np.zeros(len(tris), dtype=pt.OBJECT_DTYPE)
    o["type"] = 1
    o["material_id"] = material
    o["emission"] = emission
    o["color"] = color
    o["p1"], o["p2"], o["p3"] = tris[:, 0], tris[:, 1], tris[:, 2]
    e1, e2 = tris[:, 1] - tris[:, 0], tris[:, 2] - tris[:, 0]
    o["area"] = (0.5 * np.linalg.norm(np.cross(e1, e2), axis=1)).astype(np.float32)   # as readScene computes it
    return o


def sphere(radius, center, emission=(0, 0, 0), color=(.75, .75, .75), material=0):
    o = np.zeros(1, dtype=pt.OBJECT_DTYPE)
    o["type"], o["material_id"], o["radius"], o["center"] = 0, material, radius, center
    o["emission"], o["color"] = emission, color
    o["area"] = 4 * np.pi * radius * radius
    return o


def make_scene(objects, w, h):
    sc = pt.LoadedScene("cornell", w, h)      # camera, materials, gamma, sky of a real scene; objects replaced
    sc.objects = np.ascontiguousarray(np.concatenate(objects))
    sc.object_count = len(sc.objects)
    em = sc.objects["emission"]
    sc.light_count = int((~((em[:, 0] == 0) & (em[:, 2] == 0))).sum())
    return sc


def room():
    return [sphere(1e4, (1e4 - 54, 40.8, 81.6), color=(.75, .25, .25)), sphere(1e4, (-1e4 + 56, 40.8, 81.6), color=(.25, .25, .75)),
            sphere(1e4, (0, 40.8, 1e4)), sphere(1e4, (0, 40.8, -1e4 + 270), color=(0, 0, 0)),
            sphere(1e4, (0, 1e4, 81.6)), sphere(1e4, (0, -1e4 + 81.6, 81.6))]


def big_scene():
    """A 320-triangle emissive ball, twelve 80-triangle balls (more than 8 clusters), one 1280-triangle
    ball: 2 560 triangles, 320 lights, 11 mask words per light."""
    rng = np.random.default_rng(7)
    objs = room()
    objs.append(triangles(icosphere((-20, 30, 60), 9, 2), emission=(6, 6, 6)))
    for _ in range(12):
        objs.append(triangles(icosphere(rng.uniform((-40, 5, 30), (40, 60, 120)), rng.uniform(3, 7), 1),
                              color=rng.uniform(.3, .9, 3)))
    objs.append(triangles(icosphere((25, 15, 70), 14, 3), color=(.9, .9, .9), material=1))   # a mirror ball
    return objs


def render(sc, w, h, spp, **kw):
    g = pt.Gpu(0, 0, **kw)
    g.set_scene(sc)
    info = g.scene_info()
    g.resize(w, h, pt.seed_block(w, h))
    g.render(spp)
    out = (g.colors(), g.pixels(), g.seeds(), info)
    g.close()
    return out


def test_many_mask_words_and_clusters_strict_modes_equal_the_oracle():
    w, h, spp = 48, 36, 1
    sc = make_scene(big_scene(), w, h)
    cpu = ob.Render(sc, pt.seed_block(w, h), ob.SEEDS_KERNEL_LAYOUT, 0).pass_oracle(spp, threads=8)
    for kw in (dict(shadow_cull=pt.CULL_OFF, bvh=pt.BVH_OFF), dict(shadow_cull=pt.CULL_SAFE, bvh=pt.BVH_OFF),
               dict(bvh=pt.BVH_ON), dict(shadow_cull=pt.CULL_SAFE, bvh=pt.BVH_OFF, lds=0),
               dict(shadow_cull=pt.CULL_SAFE, bvh=pt.BVH_OFF, kernel=pt.KERNEL_MEGA)):
        colors, pixels, seeds, info = render(sc, w, h, spp, math=pt.MATH_STRICT, **kw)
        assert info["triangles"] == 2560 and info["lights"] == 320 and info["clusters"] == 8
        assert info["shadow_tests_safe"] < 0.9 * info["primitives"]
        assert (seeds == cpu.seeds).all(), kw
        assert same_bits(colors, cpu.colors), kw
        assert (pixels == cpu.pixels).all(), kw


def test_many_mask_words_and_clusters_fast_modes_agree():
    w, h, spp = 96, 72, 4
    sc = make_scene(big_scene(), w, h)
    tight = dict(shadow_cull=pt.CULL_TIGHT)
    base = render(sc, w, h, spp, math=pt.MATH_FAST, kernel=pt.KERNEL_WAVEFRONT, ray_bins=0, bvh=pt.BVH_OFF, **tight)
    for kw in (dict(ray_bins=1, bvh=pt.BVH_OFF, **tight), dict(ray_bins=1, bvh=pt.BVH_AUTO, **tight),
               dict(ray_bins=0, bvh=pt.BVH_AUTO, **tight), dict(ray_bins=1, bvh=pt.BVH_AUTO, lds=0, **tight), dict(bvh=pt.BVH_ON)):
        got = render(sc, w, h, spp, math=pt.MATH_FAST, kernel=pt.KERNEL_WAVEFRONT, **kw)
        assert (got[2] == base[2]).all(), kw
        ok = ~np.isnan(base[0]).any(1) & ~np.isnan(got[0]).any(1)
        rel = np.abs(got[0][ok] - base[0][ok]).max(1) / np.maximum(np.abs(base[0][ok]).max(1), 1e-3)
        assert (rel > 1e-4).mean() < 0.01, (kw, (rel > 1e-4).mean())   # TIGHT sets vs hierarchy: grazing rays only
    # ENCLOSED (the fast default) also leaves out the wall spheres around a light: same draws, and
    # what changes is upstream's float "acne" on the sphere a shadow ray starts from - a small share
    # of the energy (stated tolerance: 2e-3 of the mean radiance, RMSE below 1 % of it)
    for kw in (dict(ray_bins=0), dict(ray_bins=1), dict(ray_bins=1, kernel=pt.KERNEL_MEGA)):
        kw.setdefault("kernel", pt.KERNEL_WAVEFRONT)
        got = render(sc, w, h, spp, math=pt.MATH_FAST, bvh=pt.BVH_OFF, **kw)
        assert got[3]["shadow_tests_enclosed"] < got[3]["shadow_tests_tight"] - 5   # the six walls, for most lights
        assert (got[2] == base[2]).all(), kw
        ok = ~np.isnan(base[0]).any(1) & ~np.isnan(got[0]).any(1)
        d = got[0][ok].astype(np.float64) - base[0][ok]
        mean = float(base[0][ok].mean())
        print(f"enclosed vs tight {kw}: pixels that differ {(np.abs(d).max(1) > 0).mean():.3f}, "
              f"mean {d.mean() / mean:+.2e} of the radiance, rmse {np.sqrt((d ** 2).mean()) / mean:.2e}")
        assert abs(d.mean()) <= 2e-3 * mean and np.sqrt((d ** 2).mean()) <= 1e-2 * mean


def test_duplicated_and_degenerate_geometry_keeps_the_tie_rule():
    # every triangle of a ball twice (second copy with another colour: the FIRST must win every tie),
    # plus zero-area and NaN triangles and a NaN sphere, which never hit
    w, h, spp = 64, 48, 2
    ball = icosphere((0, 30, 80), 20, 2)
    junk = np.array([[[1, 1, 1], [1, 1, 1], [1, 1, 1]], [[np.nan, 0, 0], [1, 0, 0], [0, 1, 0]],
                     [[0, 0, 0], [1, 0, 0], [2, 0, 0]]], dtype=np.float32)
    objs = room() + [triangles(ball, color=(.9, .1, .1)), triangles(ball, color=(.1, .9, .1)), triangles(junk),
                     sphere(np.float32(np.nan), (0, 40, 80)), sphere(5, (0, 70, 80), emission=(12, 12, 12))]
    sc = make_scene(objs, w, h)
    cpu = ob.Render(sc, pt.seed_block(w, h), ob.SEEDS_KERNEL_LAYOUT, 0).pass_oracle(spp, threads=8)
    red, green = cpu.colors[:, 0].sum(), cpu.colors[:, 1].sum()
    assert red > green                                    # the first (red) copy is the one that is seen
    for kw in (dict(bvh=pt.BVH_OFF), dict(bvh=pt.BVH_ON), dict(bvh=pt.BVH_ON, kernel=pt.KERNEL_MEGA, lds=0)):
        colors, pixels, seeds, _ = render(sc, w, h, spp, math=pt.MATH_STRICT, **kw)
        assert (seeds == cpu.seeds).all() and same_bits(colors, cpu.colors) and (pixels == cpu.pixels).all(), kw


def test_scene_too_large_for_the_tables_falls_back_to_the_hierarchy():
    # 20 480 emissive triangles: lights x primitives = 4e8 decisions, beyond what set_scene builds
    # tables for; it must neither blow up nor change results: the BVH (AUTO) carries the scene.
    w, h = 8, 8
    sc = make_scene(room() + [triangles(icosphere((0, 40, 80), 15, 5), emission=(1, 1, 1))], w, h)
    assert sc.light_count == 20480
    with pytest.raises(pt.GpuError):
        pt.shadow_candidates(sc, pt.CULL_SAFE)
    cpu = ob.Render(sc, pt.seed_block(w, h), ob.SEEDS_KERNEL_LAYOUT, 1).pass_oracle(1, threads=8)
    g = pt.Gpu(0, pt.INTEGRATOR_DL, math=pt.MATH_STRICT)
    g.set_scene(sc)
    info = g.scene_info()
    assert info["clusters"] == 0 and info["shadow_tests_tight"] == info["primitives"] == 20486
    assert info["bvh_primitives"] == 20480
    g.resize(w, h, pt.seed_block(w, h))
    g.render(1)
    assert (g.seeds() == cpu.seeds).all() and same_bits(g.colors(), cpu.colors)
    g.close()


def _hull(points, faces):
    p = np.array(points, dtype=np.float32)
    return np.array([[p[a], p[b], p[c]] for a, b, c in faces], dtype=np.float32)


def _box(lo, hi, faces=6, dent=False):
    (x0, y0, z0), (x1, y1, z1) = lo, hi
    v = [(x0, y0, z0), (x1, y0, z0), (x1, y1, z0), (x0, y1, z0), (x0, y0, z1), (x1, y0, z1), (x1, y1, z1), (x0, y1, z1)]
    if dent:   # one corner pushed to the middle: still closed, no longer convex
        v[6] = ((x0 + x1) / 2, (y0 + y1) / 2, (z0 + z1) / 2)
    quads = [(0, 3, 2, 1), (0, 1, 5, 4), (1, 2, 6, 5), (2, 3, 7, 6), (3, 0, 4, 7), (4, 5, 6, 7)][:faces]
    return _hull(v, [t for a, b, c, d in quads for t in ((a, b, c), (a, c, d))])


def flat_hexagon(center=(0.0, 45.0, 110.0), radius=9.0):
    """A flat, double-sided hexagon, its two sides fan-triangulated from different corners: every edge is used by
    exactly two of its 8 triangles and every vertex lies in every face plane - "closed" and "convex" by those two
    tests, with no volume.  It must keep its triangles (round 2's advisor finding: planes that bound an infinite slab
    would shadow rays far outside the polygon)."""
    c = np.array(center)
    v = [c + radius * np.array((np.cos(k * np.pi / 3), np.sin(k * np.pi / 3), 0.0)) for k in range(6)]
    return _hull(v, [(0, 1, 2), (0, 2, 3), (0, 3, 4), (0, 4, 5), (1, 0, 5), (1, 5, 4), (1, 4, 3), (1, 3, 2)])


def convex_scene():
    c, r = np.array((-30.0, 12.0, 95.0)), 9.0
    octa = _hull([c + (r, 0, 0), c - (r, 0, 0), c + (0, r, 0), c - (0, r, 0), c + (0, 0, r), c - (0, 0, r)],
                 [(0, 2, 4), (2, 1, 4), (1, 3, 4), (3, 0, 4), (2, 0, 5), (1, 2, 5), (3, 1, 5), (0, 3, 5)])
    t = np.array((30.0, 4.0, 100.0))
    up = (0, 13, 0)   # a prism: two caps, three sides of two triangles each, 5 planes
    prism = _hull([t, t + (14, 0, 0), t + (7, 0, 12), t + up, t + (14, 0, 0) + up, t + (7, 0, 12) + up],
                  [(0, 1, 2), (3, 5, 4), (0, 3, 4), (0, 4, 1), (1, 4, 5), (1, 5, 2), (2, 5, 3), (2, 3, 0)])
    light = _hull([(-12, 80, 70), (12, 80, 70), (12, 80, 94), (-12, 80, 94)], [(0, 1, 2), (0, 2, 3), (0, 2, 1), (0, 3, 2)])
    objs = room() + [triangles(light, emission=(10, 10, 10)), triangles(flat_hexagon(), color=(.6, .6, .2)),
                     triangles(_box((-10, 0, 60), (10, 22, 80)), color=(.8, .6, .3)), triangles(octa, color=(.3, .8, .4)),
                     triangles(prism, color=(.4, .5, .9)), triangles(_box((20, 0, 40), (40, 15, 60), faces=5)),
                     triangles(_box((-45, 0, 40), (-25, 18, 60), dent=True), color=(.9, .3, .3))]
    return objs


def test_closed_convex_clusters_are_tested_by_their_face_planes():
    # A cube, an octahedron and a prism (closed, convex: near shadow rays test 6 / 8 / 5 planes), an open box,
    # a dented cube (closed, not convex) and a flat double-sided hexagon (closed and "convex", no volume), which
    # keep their triangles.  The plane test answers "does the
    # segment cross the boundary between EPSILON and maxt", which is what the any-hit scan over the triangles
    # answers: bins on and off must give the same image up to rays grazing an edge - and the diffuse meshes
    # start shadow rays ON themselves, the case the range (EPSILON, maxt) is there for.
    w, h, spp = 128, 96, 8
    sc = make_scene(convex_scene(), w, h)
    for cull in (pt.CULL_TIGHT, pt.CULL_ENCLOSED):
        on = render(sc, w, h, spp, math=pt.MATH_FAST, kernel=pt.KERNEL_WAVEFRONT, ray_bins=1, shadow_cull=cull)
        off = render(sc, w, h, spp, math=pt.MATH_FAST, kernel=pt.KERNEL_WAVEFRONT, ray_bins=0, shadow_cull=cull)
        info = on[3]
        assert info["clusters"] == 6 and info["convex_clusters"] == 3, info   # (the flat hexagon is the sixth)
        assert (on[2] == off[2]).all()
        ok = ~np.isnan(on[0]).any(1) & ~np.isnan(off[0]).any(1)
        d = on[0][ok].astype(np.float64) - off[0][ok]
        rel = np.abs(d).max(1) / np.maximum(np.abs(off[0][ok]).max(1), 1e-3)
        mean = float(off[0][ok].mean())
        print(f"convex planes vs triangles, cull {cull}: pixels off by > 1e-4 {(rel > 1e-4).mean():.4f}, mean {d.mean() / mean:+.2e}")
        assert (rel > 1e-4).mean() < 0.01 and abs(d.mean()) <= 2e-4 * mean
    # the strict build never takes the plane test: bit-exact against the oracle with the bins asked for
    cpu = ob.Render(sc, pt.seed_block(w, h), ob.SEEDS_KERNEL_LAYOUT, 0).pass_oracle(1, threads=8)
    got = render(sc, w, h, 1, math=pt.MATH_STRICT, kernel=pt.KERNEL_WAVEFRONT, ray_bins=1)
    assert (got[2] == cpu.seeds).all() and same_bits(got[0], cpu.colors)


def nested_room(shells):
    """`shells` opaque spheres around the whole scene (concentric, radii 2000, 2001, ...), a triangle light, two
    diffuse balls: every shell holds the light, so the ENCLOSED family may drop up to 32 of them."""
    objs = [sphere(2000.0 + k, (0, 40, 80), color=(.6, .6, .6)) for k in range(shells)]
    light = np.array([[[-10, 70, 70], [10, 70, 70], [0, 70, 90]], [[-10, 70, 70], [0, 70, 90], [-12, 70, 88]]], dtype=np.float32)
    objs += [triangles(light, emission=(30, 30, 30)),                          # normals (0, -1, 0): facing down
             sphere(12, (-15, 12, 70), color=(.8, .3, .3)), sphere(9, (18, 9, 95), color=(.3, .8, .3)),
             triangles(np.array([[[-60, 0, 20], [60, 0, 20], [60, 0, 140]], [[-60, 0, 20], [60, 0, 140], [-60, 0, 140]]], dtype=np.float32))]
    return objs


@pytest.mark.parametrize("shells", [6, 32, 40], ids=["6 shells", "32 shells", "40 shells"])
def test_enclosed_family_with_up_to_and_beyond_32_dropped_spheres_equals_its_replay(shells):
    # The kernels watch the dropped spheres as the first n_enc <= 32 slots of the shadow order; a scene with more
    # enclosing spheres than that keeps the others in the sets.  Strict build + each mode = the oracle replaying
    # the same byte matrix, both kernels; and the matrix drops exactly min(shells, 32) spheres for both lights.
    w, h, spp = 64, 48, 2
    sc = make_scene(nested_room(shells), w, h)
    seeds = pt.seed_block(w, h)
    for cull in (pt.CULL_ENCLOSED, pt.CULL_ENCLOSED_SELF, pt.CULL_BALL):
        keep = pt.shadow_candidates(sc, cull)
        dropped = ((keep == 4) | (keep == 3)).sum(1)
        assert (dropped == min(shells, 32)).all(), (cull, dropped)
        with ob.shadow_candidates(keep):
            cpu = ob.Render(sc, seeds, ob.SEEDS_KERNEL_LAYOUT, 0).pass_oracle(spp, threads=8)
        for kernel in (pt.KERNEL_MEGA, pt.KERNEL_WAVEFRONT):
            g = pt.Gpu(0, 0, math=pt.MATH_STRICT, kernel=kernel, shadow_cull=cull, bvh=pt.BVH_OFF)
            g.set_scene(sc)
            g.resize(w, h, seeds)
            g.render(spp)
            g.finish()
            assert g.launch_info()["shadow_cull"] == cull
            assert same_bits(g.colors(), cpu.colors) and (g.seeds() == cpu.seeds).all(), (shells, cull, kernel)
            g.close()
    # and the fast build's default against the strict build with the reference's own scan: no drift
    a = render(sc, 128, 96, 16, math=pt.MATH_FAST)[0]
    b = render(sc, 128, 96, 16, math=pt.MATH_STRICT, shadow_cull=pt.CULL_OFF, bvh=pt.BVH_OFF)[0]
    ok = np.isfinite(a).all(1) & np.isfinite(b).all(1)
    assert abs((a[ok].astype(np.float64) - b[ok]).mean()) <= 2e-3 * b[ok].mean()
