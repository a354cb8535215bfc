"""Which rays flip between the fast and the strict build on a hierarchy scene.

With equal RNG state in every pixel, a pixel of the fast build's image differs from the strict build's beyond rounding
only if some ray of its path got another answer: an extension ray another surface, a shadow ray the other occlusion
verdict.  The CPU oracle logs every ray of a serial render (oracle_set_ray_log: the STRICT rays - the reference's);
the ray-probe build of the library answers those same rays through the strict linear scan (sanity: must equal the
oracle's answers), the strict hierarchy, and the fast build's hierarchy instance; every ray whose fast answer is another
decision is classified from a float64 intersection of that ray with the whole scene.

usage (GPU box): flipped_rays.py [scene=table-and-mirror] [dl|pt] [width=512] [height=384]
"""
import ctypes as C
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(REPO, "oracle"))
sys.path.insert(0, os.path.join(REPO, "path-tracer_amd"))
import oracle_binding as ob
import pyptgpu as pt

EPS = 0.01


def oracle_rays(sc, seeds, integrator, capacity):
    lib = ob.oracle()
    lib.oracle_set_ray_log.argtypes = [C.c_void_p, C.c_long]
    lib.oracle_set_ray_log.restype = None
    lib.oracle_ray_count.restype = C.c_long
    log = np.zeros((capacity, 12), np.float32)
    lib.oracle_set_ray_log(log.ctypes.data_as(C.c_void_p), capacity)
    try:
        img = ob.Render(sc, seeds, ob.SEEDS_KERNEL_LAYOUT, integrator).pass_oracle(1, threads=1)
        n = lib.oracle_ray_count()
    finally:
        lib.oracle_set_ray_log(None, 0)
    assert n < capacity, "ray log full"
    return log[:n].copy(), img


def exact_hits(sc, o, d):
    """float64 intersections of ONE ray with every primitive: rows (t, primitive, edge distance in barycentric units
    or -1 for a sphere), t > 0, sorted by t."""
    objs = sc.objects
    o = o.astype(np.float64); d = d.astype(np.float64)
    rows = []
    tri = np.nonzero(objs["type"] == 1)[0]
    if len(tri):
        p1, p2, p3 = (objs[k][tri].astype(np.float64) for k in ("p1", "p2", "p3"))
        e1, e2 = p2 - p1, p3 - p1
        pv = np.cross(d[None, :], e2)
        det = (e1 * pv).sum(1)
        ok = np.abs(det) > 0
        inv = np.where(ok, 1.0 / np.where(ok, det, 1), 0)
        tv = o[None, :] - p1
        u = (tv * pv).sum(1) * inv
        qv = np.cross(tv, e1)
        v = (qv * d[None, :]).sum(1) * inv
        t = (e2 * qv).sum(1) * inv
        edge = np.minimum(np.minimum(u, v), 1 - u - v)
        near = ok & (t > 0) & (edge > -1e-3)          # inside, or within 1e-3 (barycentric) outside an edge
        for i in np.nonzero(near)[0]:
            rows.append((t[i], int(tri[i]), edge[i]))
    sph = np.nonzero(objs["type"] == 0)[0]
    for i in sph:
        c, r = objs["center"][i].astype(np.float64), float(objs["radius"][i])
        op = c - o
        b = op @ d
        disc = b * b - op @ op + r * r
        if disc >= 0:
            s = np.sqrt(disc)
            for t in (b - s, b + s):
                if t > 0:
                    rows.append((t, int(i), -1.0))
    rows.sort()
    return rows


def classify(sc, rec, strict_answer, fast_answer):
    """One flipped ray -> a class name."""
    o, d, maxt = rec[1:4], rec[4:7], float(rec[7])
    hits = exact_hits(sc, o, d)
    shadow = rec[0] == 1
    limit = maxt if shadow else 1e30
    inside = [h for h in hits if 0.5 * EPS <= h[0] < limit * (1 + 1e-5)]   # (closer than EPSILON: never a hit, geomfunc.h:36)
    classes = []
    for t, prim, edge in inside:
        if t <= 2 * EPS:
            classes.append("start: a surface within 2 EPSILON of the origin (the ray's own surface or its neighbour)")
        if shadow and abs(t - maxt) <= 2e-4 * max(maxt, 1.0):
            classes.append("end: a surface within 2e-4 of maxt (the light's own neighbourhood)")
        if -1e-3 < edge < 1e-4:
            classes.append("edge: a triangle crossed within 1e-4 (barycentric) of an edge or vertex")
    if not classes and not shadow and len(inside) >= 2 and abs(inside[0][0] - inside[1][0]) <= 1e-5 * max(inside[0][0], 1.0):
        classes.append("tie: two surfaces at distances within 1e-5 relative")
    if not classes and not shadow and strict_answer > 0 and fast_answer > 0 and abs(strict_answer - fast_answer) <= 1e-3 * strict_answer:
        classes.append("same surface region, distance differs > 1e-4 relative (grazing incidence)")
    order = ("edge", "start", "end", "tie", "same")
    for key in order:
        for c in classes:
            if c.startswith(key):
                return c
    return "other"


def main():
    args = sys.argv[1:]
    scene = args[0] if args else "table-and-mirror"
    integ = 1 if (len(args) < 2 or args[1] == "dl") else 0
    w = int(args[2]) if len(args) > 2 else 512
    h = int(args[3]) if len(args) > 3 else 384
    sc = pt.LoadedScene(scene, w, h)
    seeds = pt.seed_block(w, h)

    def render(math, **kw):
        g = pt.Gpu(0, integ, math=math, **kw); g.set_scene(sc); g.resize(w, h, seeds); g.render(1); g.finish()
        out = (g.colors(), g.seeds(), g.launch_info()); g.close()
        return out

    fast, strict = render(pt.MATH_FAST), render(pt.MATH_STRICT)
    log, img = oracle_rays(sc, seeds, integ, 12 * w * h)
    assert img.colors.tobytes() == strict[0].tobytes(), "strict build != oracle"
    a, b = fast[0].astype(np.float64), strict[0].astype(np.float64)
    ok = np.isfinite(a).all(1) & np.isfinite(b).all(1)
    rel = np.where(ok, np.abs(a - b).max(1) / np.maximum(np.abs(b).max(1), 1e-3), 0)
    # colours are stored bottom-up: slot = (h - 1 - y) * w + x; the ray log names pixels y * w + x
    diff_slots = np.nonzero(rel > 1e-3)[0]
    diff_pixels = set(int((h - 1 - s // w) * w + s % w) for s in diff_slots)
    mean = b[ok].mean()
    print(f"{scene} {'dl' if integ else 'pt'} {w}x{h}x1: fast instance {fast[2]['instance']} kernel {fast[2]['kernel']}; seeds equal "
          f"{(fast[1] == strict[1]).all(1).mean() if fast[1].ndim > 1 else (fast[1] == strict[1]).mean():.5f}; pixels off by > 1e-3: {len(diff_slots)} of {w * h} "
          f"({len(diff_slots) / (w * h):.4%}); rel rmse {np.sqrt(((a - b)[ok] ** 2).mean()) / mean:.4f}, mean bias {(a - b)[ok].mean() / mean:+.2e}")

    o, d, maxt = log[:, 1:4], log[:, 4:7], log[:, 7]
    def probe(math, **kw):
        g = pt.Gpu(0, integ, lib=pt.variant_path("probe"), math=math, **kw); g.set_scene(sc); g.resize(8, 8, pt.seed_block(8, 8))
        t, rgb, occ = g.probe_rays(o, d, maxt); li = g.scene_info(); g.close()
        return t, rgb, occ
    lin = probe(pt.MATH_STRICT, bvh=pt.BVH_OFF, shadow_cull=pt.CULL_OFF)
    tree = probe(pt.MATH_STRICT, bvh=pt.BVH_ON)
    fst = probe(pt.MATH_FAST)
    ext, sh = log[:, 0] == 0, log[:, 0] == 1
    ans = log[:, 8]
    lin_t = np.where(lin[0] < 0, 0, lin[0])
    print(f"rays logged: {ext.sum()} extension, {sh.sum()} shadow; strict linear scan on the GPU against the oracle: "
          f"{(lin_t[ext].view(np.uint32) != ans[ext].view(np.uint32)).sum()} distances differ, {(lin[2][sh] != (ans[sh] != 0)).sum()} occlusion answers differ; "
          f"strict hierarchy against strict linear: {(tree[0].view(np.uint32) != lin[0].view(np.uint32)).sum()} / {(tree[2] != lin[2]).sum()}")
    f_t = np.where(fst[0] < 0, 0, fst[0])
    surface_differs = ext & ((np.abs(fst[1] - lin[1]).max(1) > 1e-6) | ((f_t == 0) != (lin_t == 0)))
    far_off = ext & ~surface_differs & (np.abs(f_t - lin_t) > 1e-4 * np.maximum(lin_t, 1.0))
    occ_flip = sh & (fst[2] != lin[2])
    print(f"fast build: extension rays with another surface {surface_differs.sum()} ({surface_differs.sum() / ext.sum():.2e}), same surface but distance off by "
          f"> 1e-4 rel {far_off.sum()}, distance bits differ {(ext & (f_t.view(np.uint32) != lin_t.view(np.uint32))).sum()}; "
          f"shadow rays with the other verdict {occ_flip.sum()} ({occ_flip.sum() / max(sh.sum(), 1):.2e}; {(occ_flip & lin[2]).sum()} occluded -> free, "
          f"{(occ_flip & ~lin[2]).sum()} free -> occluded)")
    flipped = np.nonzero(surface_differs | far_off | occ_flip)[0]
    table = {}
    for i in flipped[:4000]:
        kind = "shadow" if log[i, 0] == 1 else "extension"
        c = classify(sc, log[i], float(lin_t[i]), float(f_t[i]))
        table[(kind, c)] = table.get((kind, c), 0) + 1
    print("| ray | class | rays |\n|---|---|---|")
    for (kind, c), n in sorted(table.items(), key=lambda kv: -kv[1]):
        print(f"| {kind} | {c} | {n} |")
    flipped_pixels = set(int(p) for p in log[flipped, 10])
    print(f"pixels off by > 1e-3: {len(diff_pixels)}; of them with a flipped (strict) ray on their path: {len(diff_pixels & flipped_pixels)}; "
          f"pixels with a flipped ray whose colour did not move: {len(flipped_pixels - diff_pixels)}")
    # the others: paths whose rays differ by rounding only - how big are their pixel differences?
    rest = sorted(diff_pixels - flipped_pixels)
    if rest:
        slots = [(h - 1 - p // w) * w + p % w for p in rest]
        depth_of = {}
        for p, dep in zip(log[:, 10].astype(int), log[:, 11].astype(int)):
            depth_of[p] = max(depth_of.get(p, 0), dep)
        depths = np.array([depth_of.get(p, -1) for p in rest])
        print(f"the other {len(rest)}: rel difference median {np.median(rel[slots]):.2e}, max {rel[slots].max():.2e}; path segments: "
              + ", ".join(f"{k}: {int((depths == k).sum())}" for k in sorted(set(depths.tolist()))))
        pick = sorted(rest, key=lambda p: -rel[(h - 1 - p // w) * w + p % w])[:4] + rest[:: max(1, len(rest) // 4)][:4]
        for p in pick:
            slot = (h - 1 - p // w) * w + p % w
            rays = log[log[:, 10] == p]
            first = rays[0]
            obj = sc.objects[int(first[9])] if first[9] >= 0 else None
            hit = first[1:4] + first[8] * first[4:7]
            print(f"  pixel ({p % w}, {p // w}): fast {a[slot]} strict {b[slot]}; first hit object {int(first[9])} "
                  f"({'sphere r ' + str(obj['radius']) if obj is not None and obj['type'] == 0 else 'triangle'}) at t {first[8]:.4f}, hit {hit}; "
                  f"shadow rays (maxt, occluder, its type/radius, its t) "
                  f"{[(float(r[7]), int(r[8]) - 1, ('sphere r %g' % sc.objects['radius'][int(r[8]) - 1] if sc.objects['type'][int(r[8]) - 1] == 0 else 'triangle') if r[8] > 0 else None, float(r[9])) for r in rays if r[0] == 1]}")
        # which materials do those paths touch?  (a specular bounce magnifies the rounding of its direction)
        mats = {}
        ids_of = {}
        for r in log[(log[:, 0] == 0)]:
            ids_of.setdefault(int(r[10]), []).append(int(r[9]))
        for p in rest:
            key = tuple(int(sc.materials["type"][sc.objects["material_id"][i]]) if i >= 0 else -1 for i in ids_of.get(p, []))
            mats[key] = mats.get(key, 0) + 1
        print("material types along those paths (0 lambertian, 1 conductor, 2 dielectric, 3 matte ...; -1 miss):",
              sorted(mats.items(), key=lambda kv: -kv[1])[:8])


if __name__ == "__main__":
    main()
