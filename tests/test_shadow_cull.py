"""Shadow-ray candidate sets (PTGPU_OPT_SHADOW_CULL): the host-side tables and, replayed by the CPU
oracle, their effect on the image.  No GPU: ptgpu_shadow_candidates is host code in libptgpu.so.

What is proven here on the CPU
  * SAFE never changes a bit of the oracle's image (colours, pixels, RNG state) on the reference's
    scenes, both integrators - the same bar as the brute-force scan;
  * TIGHT is a subset of SAFE, culls only primitives that lie in the closed half-space behind the
    light's plane, and moves the image by less than the calibrated FP floor (it can differ only for
    rays that graze the light's own plane, where the contribution's cosine factor is ~0);
  * ENCLOSED is TIGHT minus the opaque spheres a light lies wholly inside - for the rays that start
    inside those spheres (keep value 4: every path vertex is checked through the extension ray that
    found it, upstream's paths do leave their room through its corners) - never drops a dielectric
    sphere or a sphere the light pokes out of, and moves the image by 1e-6 .. 1e-5 of its radiance,
    no sign (upstream's float "acne" on the sphere a shadow ray starts from); round 2's version
    assumed the paths stay inside and was brighter than upstream by 1.1e-4, one-signed;
  * ENCLOSED_SELF (keep value 3: also the rays that start ON a dropped sphere test it) is TIGHT, bit
    for bit.
"""
import numpy as np
import pytest

import oracle_binding as ob
import pyptgpu as pt
from conftest import same_bits

SCENES = ["cornell", "materials", "cornell-with-hole", "obj-model", "simple", "cornell-large", "caustic",
          "table-and-mirror", "stacked-spheres"]


def geometry(sc):
    o = sc.objects
    real = (o["type"] == 0) | (o["type"] == 1)
    emissive = ~((o["emission"][:, 0] == 0) & (o["emission"][:, 2] == 0)) & real   # vec.h:23 looks at x and z
    lights = np.nonzero(emissive)[0][: sc.light_count]
    return o, real, lights


@pytest.mark.parametrize("name", SCENES)
def test_tables_are_consistent(name):
    sc = pt.LoadedScene(name, 32, 24)
    o, real, lights = geometry(sc)
    off = pt.shadow_candidates(sc, pt.CULL_OFF)
    safe = pt.shadow_candidates(sc, pt.CULL_SAFE)
    tight = pt.shadow_candidates(sc, pt.CULL_TIGHT)
    assert off.shape == safe.shape == tight.shape == (len(lights), sc.object_count)
    assert (off == real[None, :]).all()                      # OFF: every sphere/triangle, no MODEL placeholder
    assert (safe <= off).all() and (tight <= safe).all()     # nested
    for l, li in enumerate(lights):
        if o["type"][li] == 0:                               # sphere lights are sampled from every side
            assert (safe[l] == off[l]).all() and (tight[l] == off[l]).all()
            continue
        assert safe[l, li] == 1                              # a light is in front of nothing, itself included
        p0 = o["p1"][li].astype(np.float64)
        n = np.cross(o["p2"][li].astype(np.float64) - p0, o["p3"][li].astype(np.float64) - p0)
        n /= np.linalg.norm(n)
        for i in np.nonzero(real & (tight[l] == 0))[0]:      # whatever TIGHT culls lies behind the plane
            if o["type"][i] == 1:
                top = max(float((o[k][i].astype(np.float64) - p0) @ n) for k in ("p1", "p2", "p3"))
                scale = max(np.linalg.norm(o[k][i].astype(np.float64) - p0) for k in ("p1", "p2", "p3"))
            else:
                top = float((o["center"][i].astype(np.float64) - p0) @ n) + abs(float(o["radius"][i]))
                scale = np.linalg.norm(o["center"][i].astype(np.float64) - p0) + abs(float(o["radius"][i]))
            assert top <= 1e-4 * (scale + 1.0), (name, l, i, top)
        for i in np.nonzero(real & (safe[l] == 0))[0]:       # SAFE: behind by at least 10 EPSILON
            if o["type"][i] == 1:
                top = max(float((o[k][i].astype(np.float64) - p0) @ n) for k in ("p1", "p2", "p3"))
            else:
                top = float((o["center"][i].astype(np.float64) - p0) @ n) + abs(float(o["radius"][i]))
            assert top < -0.1


def test_light_count_limits_rows_and_bad_arguments():
    sc = pt.LoadedScene("cornell", 32, 24)
    assert pt.shadow_candidates(sc, pt.CULL_SAFE, light_count=5).shape == (5, sc.object_count)
    assert pt.shadow_candidates(sc, pt.CULL_SAFE, light_count=0).shape == (0, sc.object_count)
    with pytest.raises(pt.GpuError):
        pt.shadow_candidates(sc, 7)


@pytest.mark.parametrize("integrator", [0, 1], ids=["pt", "dl"])
@pytest.mark.parametrize("case", [("cornell", 64, 48, 3), ("materials", 96, 64, 4), ("cornell-with-hole", 48, 36, 3),
                                  ("obj-model", 40, 30, 2), ("table-and-mirror", 48, 36, 3)])
def test_safe_replay_is_bit_identical_on_the_cpu(case, integrator):
    name, w, h, spp = case
    sc = pt.LoadedScene(name, w, h)
    seeds = pt.seed_block(w, h)
    full = ob.Render(sc, seeds, ob.SEEDS_KERNEL_LAYOUT, integrator).pass_oracle(spp, threads=8)
    with ob.shadow_candidates(pt.shadow_candidates(sc, pt.CULL_SAFE)):
        culled = ob.Render(sc, seeds, ob.SEEDS_KERNEL_LAYOUT, integrator).pass_oracle(spp, threads=8)
    assert same_bits(culled.colors, full.colors)
    assert (culled.pixels == full.pixels).all() and (culled.seeds == full.seeds).all()


def test_tight_replay_moves_cornell_less_than_the_fp_floor():
    # two CPU builds of the reference that differ only in FP evaluation disagree on
    # cornell 48x36x8 with RMSE 1.1e-2 and on 51.7 % of the pixels; TIGHT must stay inside that.
    w, h, spp = 48, 36, 8
    sc = pt.LoadedScene("cornell", w, h)
    seeds = pt.seed_block(w, h)
    full = ob.Render(sc, seeds).pass_oracle(spp, threads=8)
    with ob.shadow_candidates(pt.shadow_candidates(sc, pt.CULL_TIGHT)):
        culled = ob.Render(sc, seeds).pass_oracle(spp, threads=8)
    assert (culled.seeds == full.seeds).all()                 # visibility never changes the draw sequence
    ok = ~np.isnan(full.colors).any(1) & ~np.isnan(culled.colors).any(1)
    assert (np.isnan(full.colors).any(1) == np.isnan(culled.colors).any(1)).all()
    d = (culled.colors[ok].astype(np.float64) - full.colors[ok])
    rmse = np.sqrt((d ** 2).mean())
    print('TIGHT replay: rmse', rmse, 'share moved', (np.abs(d).max(1) > 1e-4).mean())
    assert rmse <= 1.1e-2
    assert (np.abs(d).max(1) > 1e-4).mean() <= 0.1 * 0.517


def test_counting_replay_reports_fewer_tests_and_same_draws():
    sc = pt.LoadedScene("cornell", 32, 24)
    seeds = pt.seed_block(32, 24)
    a, b = ob.Counters(), ob.Counters()
    ob.Render(sc, seeds).pass_oracle_counted(a, 1)
    with ob.shadow_candidates(pt.shadow_candidates(sc, pt.CULL_SAFE)):
        ob.Render(sc, seeds).pass_oracle_counted(b, 1)
    assert (a.rng_draws, a.light_samples, a.extension_rays, a.shadow_rays) == \
           (b.rng_draws, b.light_samples, b.extension_rays, b.shadow_rays)
    assert b.triangle_tests < 0.8 * a.triangle_tests


def test_enclosed_drops_only_opaque_spheres_around_the_light():
    sc = pt.LoadedScene("cornell", 32, 24)
    o, real, lights = geometry(sc)
    tight = pt.shadow_candidates(sc, pt.CULL_TIGHT)
    enc = pt.shadow_candidates(sc, pt.CULL_ENCLOSED)
    # 4: a dropped sphere stays for the rays of a vertex that may be outside one - whatever the light, also where TIGHT
    # itself culls the sphere (the kernels keep one bit per vertex and sphere, not per light; testing more than a
    # set asks for is upstream's own scan)
    assert set(np.unique(enc)) <= {0, 1, 4} and ((enc == 1) <= (tight == 1)).all()
    assert (o["type"][np.argwhere(enc == 4)[:, 1]] == 0).all()
    changed = np.argwhere((tight == 1) & (enc != 1))
    assert len(changed) > 500                                   # ~5 of the 6 wall spheres for each of 123 lights
    for l, i in changed:
        li = lights[l]
        assert o["type"][i] == 0 and o["type"][li] == 1 and o["material_id"][i] != 2
        c, r = o["center"][i].astype(np.float64), float(o["radius"][i])
        for v in (o["p1"][li], o["p2"][li], o["p3"][li]):
            assert np.linalg.norm(v.astype(np.float64) - c) <= r * (1 - 1e-5)
    # the cylinder stands in the floor (its lower rim is at y = -8, the floor sphere's top at y = 0):
    # the lights that reach below it keep the floor sphere, and nothing else is kept of the walls
    walls = np.nonzero((o["type"] == 0) & (o["radius"] >= 1000))[0]
    floor = [i for i in walls if o["center"][i][1] > 9000][0]
    kept = (enc[:, walls] == 1).sum(0)
    assert kept[list(walls).index(floor)] > 30 and kept.sum() == kept[list(walls).index(floor)]
    # a glass wall encloses nothing: paths leave through it
    sc.objects["material_id"][walls] = 2
    glass = pt.shadow_candidates(sc, pt.CULL_ENCLOSED)
    assert (glass == tight).all()


@pytest.mark.parametrize("name", ["cornell", "cornell-large", "cornell-original"])
def test_enclosed_family_replay_against_tight(name):
    # ENCLOSED_SELF answers every shadow ray as TIGHT does: the same image, bit for bit.  ENCLOSED leaves out the
    # sphere a ray STARTS on: grazing rays whose weight is ~0 - some pixels move, the mean does not (<= 2e-5 of the
    # radiance; round 2's ENCLOSED, which assumed that paths never leave the spheres, was +1.1e-4 .. +1.3e-4 here).
    # The premise that fails upstream is visible in the oracle's own log: shadow rays that start OUTSIDE a wall.
    w, h, spp = 96, 72, 6
    sc = pt.LoadedScene(name, w, h)
    seeds = pt.seed_block(w, h)
    res = {}
    for mode in (pt.CULL_TIGHT, pt.CULL_ENCLOSED, pt.CULL_ENCLOSED_SELF):
        with ob.shadow_candidates(pt.shadow_candidates(sc, mode)):
            res[mode] = ob.Render(sc, seeds, ob.SEEDS_KERNEL_LAYOUT, 0).pass_oracle(spp, threads=8)
    tight, enc, own = res[pt.CULL_TIGHT], res[pt.CULL_ENCLOSED], res[pt.CULL_ENCLOSED_SELF]
    assert (tight.seeds == enc.seeds).all() and (tight.seeds == own.seeds).all()   # draws never depend on a shadow test
    assert same_bits(own.colors, tight.colors) and (own.pixels == tight.pixels).all()
    ok = ~(np.isnan(tight.colors).any(1) | np.isnan(enc.colors).any(1))
    d = enc.colors[ok].astype(np.float64) - tight.colors[ok]
    mean = float(tight.colors[ok].mean())
    print(f"{name}: ENCLOSED vs TIGHT on the CPU: {(np.abs(d).max(1) > 0).mean():.3f} of the pixels differ, "
          f"mean {d.mean() / mean:+.2e}, rmse {np.sqrt((d ** 2).mean()) / mean:.2e} of the radiance")
    assert (d >= -1e-6).all()                                   # only occlusions are lost: nothing gets darker
    assert 0 <= d.mean() <= 2e-5 * mean and np.sqrt((d ** 2).mean()) <= 1e-2 * mean


def test_paths_leave_the_room_through_its_corners_and_enclosed_keeps_their_walls():
    # The finding behind round 3's ENCLOSED: with the sets of a rule that assumes "a path never leaves an opaque
    # sphere" (4 -> 0: round 2's ENCLOSED), the oracle's mismatch log shows shadow rays that the full scan finds
    # occluded by a wall the origin is OUTSIDE of (a bounce ray that starts within EPSILON of a second wall passes
    # through it, geomfunc.h:36) - every light of such a vertex, and that is the one-signed bias.  With the
    # per-vertex check (4: did the ray that found the vertex get every dropped sphere's far root?) the mismatches
    # left are rays occluded by the sphere they start on - or, within rounding of a corner, by its neighbour.
    import ctypes as C
    lib = ob.oracle()
    lib.oracle_set_mismatch_log.argtypes = [C.c_void_p, C.c_int]
    lib.oracle_set_mismatch_log.restype = None
    w, h, spp = 96, 72, 2
    sc = pt.LoadedScene("cornell", w, h)
    o = sc.objects
    seeds = pt.seed_block(w, h)
    tight = pt.shadow_candidates(sc, pt.CULL_TIGHT)
    enc = pt.shadow_candidates(sc, pt.CULL_ENCLOSED)
    naive = np.where(enc == 4, 0, enc).astype(np.uint8)

    def mismatches(keep):
        log = np.zeros((100000, 12), dtype=np.float32)
        lib.oracle_set_mismatch_log(log.ctypes.data, len(log))
        try:
            with ob.shadow_candidates(keep):
                ob.Render(sc, seeds, ob.SEEDS_KERNEL_LAYOUT, 0).pass_oracle(spp, threads=1)
            n = lib.oracle_mismatch_count()
        finally:
            lib.oracle_set_mismatch_log(None, 0)
        rec = log[:n]
        spheres = o["type"][rec[:, 9].astype(int)] == 0        # the full scan's occluder is a sphere (else: TIGHT's class)
        return rec[spheres]

    base = len(mismatches(tight))                                # (TIGHT never drops a sphere: nothing to find)
    assert base == 0
    lost = mismatches(naive)
    own = lost[:, 8] == lost[:, 9]
    outside = []
    for r in lost[~own]:
        c, rad = o["center"][int(r[9])].astype(np.float64), float(o["radius"][int(r[9])])
        outside.append(np.linalg.norm(r[:3].astype(np.float64) - c) - rad)
    outside = np.array(outside)
    print(f"assumed-inside sets: {len(lost)} shadow rays lose a sphere occluder; {own.sum()} start on it, "
          f"{(~own).sum()} elsewhere, of which {(outside > 0).sum()} start outside that sphere (median {np.median(outside):+.3f})")
    assert (~own).sum() > 500 and (outside > -0.01).all() and (outside > 0).mean() > 0.9
    kept = mismatches(enc)
    depth = np.array([float(o["radius"][int(r[9])]) - np.linalg.norm(r[:3].astype(np.float64) - o["center"][int(r[9])].astype(np.float64))
                      for r in kept])
    assert len(kept) > 0 and (np.abs(depth) < 0.01).all()        # what is left: acne, origins within rounding of the sphere
    assert (kept[:, 8] == kept[:, 9]).mean() > 0.99


def test_ball_marks_the_enclosing_spheres_a_light_reaches_out_of():
    sc = pt.LoadedScene("cornell", 32, 24)
    o, real, lights = geometry(sc)
    enc = pt.shadow_candidates(sc, pt.CULL_ENCLOSED)
    ball = pt.shadow_candidates(sc, pt.CULL_BALL)
    assert ((ball != 0) == (enc != 0)).all() and ((ball == 4) == (enc == 4)).all()     # the same pairs, some tested another way
    marked = np.argwhere(ball == 2)
    walls = np.nonzero((o["type"] == 0) & (o["radius"] >= 1000))[0]
    floor = [i for i in walls if o["center"][i][1] > 9000][0]
    # cornell: the floor, for every light of the cylinder that still had it
    assert len(marked) > 30 and (marked[:, 1] == floor).all() and (ball[:, floor] != 1).all()
    for l, i in marked:
        c, r = o["center"][i].astype(np.float64), float(o["radius"][i])
        inside = [np.linalg.norm(v.astype(np.float64) - c) <= r * (1 - 1e-5) for v in (o["p1"][lights[l]], o["p2"][lights[l]], o["p3"][lights[l]])]
        assert not all(inside)                                  # a light wholly inside is dropped, not marked
    # a sphere that encloses no light (the small ones of materials.txt) is never marked, sphere lights
    # mark nothing, and a scene of more than 32 spheres falls back to ENCLOSED
    for name in ("materials", "simple", "complex"):
        s2 = pt.LoadedScene(name, 32, 24)
        b2, e2 = pt.shadow_candidates(s2, pt.CULL_BALL), pt.shadow_candidates(s2, pt.CULL_ENCLOSED)
        o2 = s2.objects
        for l, i in np.argwhere(b2 == 2):
            assert o2["type"][i] == 0 and o2["material_id"][i] != 2 and (e2[:, i] == 4).any()
        if name == "complex":
            assert (b2 == e2).all()
    sc.objects["material_id"][walls] = 2                        # glass walls enclose nothing
    assert (pt.shadow_candidates(sc, pt.CULL_BALL) == pt.shadow_candidates(sc, pt.CULL_TIGHT)).all()


def test_ball_replay_moves_cornell_by_its_acne_only():
    w, h, spp = 96, 72, 6
    sc = pt.LoadedScene("cornell", w, h)
    seeds = pt.seed_block(w, h)
    with ob.shadow_candidates(pt.shadow_candidates(sc, pt.CULL_ENCLOSED)):
        enc = ob.Render(sc, seeds, ob.SEEDS_KERNEL_LAYOUT, 0).pass_oracle(spp, threads=8)
    with ob.shadow_candidates(pt.shadow_candidates(sc, pt.CULL_BALL)):
        ball = ob.Render(sc, seeds, ob.SEEDS_KERNEL_LAYOUT, 0).pass_oracle(spp, threads=8)
    assert (enc.seeds == ball.seeds).all()                      # draws never depend on a shadow test
    ok = ~(np.isnan(enc.colors).any(1) | np.isnan(ball.colors).any(1))
    d = ball.colors[ok].astype(np.float64) - enc.colors[ok]
    mean = float(enc.colors[ok].mean())
    print(f"BALL vs ENCLOSED on the CPU: {(np.abs(d).max(1) > 0).mean():.3f} of the pixels differ, "
          f"mean {d.mean() / mean:+.2e}, rmse {np.sqrt((d ** 2).mean()) / mean:.2e} of the radiance, "
          f"{(d.min(1) < -1e-6).mean():.2e} of the pixels darker")
    # the floor's self-shadowing acne of the grazing rays towards the foot of the cylinder goes
    # (brighter: +5e-4, more than ENCLOSED's, which is why BALL is opt-in); a ray end within
    # rounding of the floor's surface may be classed either way (0.5 % of the pixels get darker)
    assert (d.min(1) < -1e-6).mean() <= 1e-2
    assert 0 <= d.mean() <= 1e-3 * mean and np.sqrt((d ** 2).mean()) <= 1e-2 * mean
