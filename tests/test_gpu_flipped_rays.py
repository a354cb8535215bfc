"""Fast against strict on a hierarchy scene, ray by ray and as an image (tolerance 3).

table-and-mirror.txt under direct lighting: every pixel's RNG state is equal in both builds, yet a few pixels differ by
their whole value.  tests/tools/flipped_rays.py shows which rays: the oracle logs the reference's rays, the ray-probe
build answers them.  Through the fast hierarchy instance the SAME rays get the same answers (0 extension rays with
another surface, <= 5e-5 of the shadow rays with the other verdict); what flips in a render are shadow rays whose
verdict hangs on a comparison against EPSILON that float32 cannot resolve - a ray that starts on an r = 1e4 wall
sphere and re-hits it at t = 0.0108 (EPSILON = 0.01; the sphere's t has a resolution of 1e-3 there), or a sample on
the light's limb that the light itself occludes at t = maxt - 2e-5 - and the fast build's rays differ from the
reference's in the last bits of their direction (v_sin / v_cos / v_rsq in the light sample).  The reference's own GPU
and CPU builds differ the same way.  This file gates both levels."""
import ctypes as C
import os
import sys

import numpy as np
import pytest

import oracle_binding as ob
import pyptgpu as pt

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "tools"))

pytestmark = pytest.mark.gpu


def test_the_fast_hierarchy_instance_answers_the_references_rays_as_the_strict_scan_does():
    import flipped_rays as fr
    w, h = 256, 192
    for scene in ("table-and-mirror", "complex"):
        sc = pt.LoadedScene(scene, w, h)
        log, _ = fr.oracle_rays(sc, pt.seed_block(w, h), 1, 12 * w * h)
        o, d, maxt = log[:, 1:4], log[:, 4:7], log[:, 7]
        answers = {}
        for name, kw in (("strict", dict(math=pt.MATH_STRICT, bvh=pt.BVH_OFF, shadow_cull=pt.CULL_OFF)), ("fast", dict(math=pt.MATH_FAST))):
            g = pt.Gpu(0, 1, lib=pt.variant_path("probe"), **kw)
            g.set_scene(sc)
            g.resize(8, 8, pt.seed_block(8, 8))
            answers[name] = g.probe_rays(o, d, maxt)
            assert name == "strict" or g.scene_info()["bvh_nodes"] > 0
            g.close()
        ext, sh = log[:, 0] == 0, log[:, 0] == 1
        st, fa = answers["strict"], answers["fast"]
        # the strict scan on the GPU is the oracle's, bit for bit
        assert (np.where(st[0] < 0, 0, st[0])[ext].view(np.uint32) == log[ext, 8].view(np.uint32)).all()
        assert (st[2][sh] == (log[sh, 8] != 0)).all()
        other_surface = ext & ((np.abs(fa[1] - st[1]).max(1) > 1e-6) | ((fa[0] < 0) != (st[0] < 0)))
        far_off = ext & ~other_surface & (np.abs(fa[0] - st[0]) > 1e-4 * np.maximum(st[0], 1.0))
        flips = sh & (fa[2] != st[2])
        print(f"{scene}: {ext.sum()} extension rays, {other_surface.sum()} with another surface, {far_off.sum()} off by > 1e-4; "
              f"{sh.sum()} shadow rays, {flips.sum()} with the other verdict")
        assert other_surface.sum() <= 5e-5 * ext.sum() + 1 and far_off.sum() == 0
        assert flips.sum() <= 1e-4 * sh.sum() + 1


def test_hierarchy_scene_fast_image_against_strict_at_a_size_that_means_something():
    """table-and-mirror.txt, direct lighting, 256x192x32 (1.6 M paths).  Stated bounds: |mean bias| <= 3e-4 of the mean
    radiance (measured +3.0e-5; a flipped acne pixel near the light carries up to 4x the mean), relative RMSE <= 5 %
    (measured 2.3 %), 8x8-block relative error: median <= 0.01 % (measured 4e-7), p99 <= 12 % (measured 5.4 %, max 10 %:
    the blocks of the ceiling around the light, where one flipped pixel moves its block), equal RNG state in >= 99.99 % of the pixels, NaN pixels equal."""
    w, h, spp = 256, 192, 32
    sc = pt.LoadedScene("table-and-mirror", w, h)
    seeds = pt.seed_block(w, h)
    out = {}
    for math in (pt.MATH_FAST, pt.MATH_STRICT):
        g = pt.Gpu(0, 1, math=math)
        g.set_scene(sc)
        g.resize(w, h, seeds)
        g.render(spp)
        g.finish()
        out[math] = (g.colors(), g.seeds(), g.launch_info())
        g.close()
    (a, sa, la), (b, sb, _) = out[pt.MATH_FAST], out[pt.MATH_STRICT]
    assert la["instance"] == 1                                  # the hierarchy instance
    same_state = float((sa.reshape(-1, 2) == sb.reshape(-1, 2)).all(1).mean())   # (a roulette at the glass sphere may flip: < 1e-4)
    assert same_state >= 0.9999
    ok = np.isfinite(a).all(1) & np.isfinite(b).all(1)
    assert (~np.isfinite(a).all(1)).sum() == (~np.isfinite(b).all(1)).sum()
    d = a[ok].astype(np.float64) - b[ok]
    mean = b[ok].astype(np.float64).mean()
    blocks_a = np.where(ok[:, None], a, 0).reshape(h // 8, 8, w // 8, 8, 3).astype(np.float64).mean((1, 3, 4))
    blocks_b = np.where(ok[:, None], b, 0).reshape(h // 8, 8, w // 8, 8, 3).astype(np.float64).mean((1, 3, 4))
    blocks = np.abs(blocks_a - blocks_b) / np.maximum(blocks_b, 1e-3)
    lit_only_fast = int(((a.max(1) > 0) & (b.max(1) == 0)).sum())
    print(f"table-and-mirror dl {w}x{h}x{spp}: bias {d.mean() / mean:+.2e}, rel rmse {np.sqrt((d ** 2).mean()) / mean:.4f}, blocks median "
          f"{np.median(blocks):.2e} p99 {np.quantile(blocks, .99):.4f} max {blocks.max():.3f}; pixels off by > 1e-3: "
          f"{(np.abs(d).max(1) > 1e-3 * np.maximum(np.abs(b[ok]).max(1), 1e-3)).sum()}, brighter in fast {(d.sum(1) > 0).sum()}, darker {(d.sum(1) < 0).sum()}; "
          f"lit only in fast {lit_only_fast}; RNG state equal in {same_state:.6f} of the pixels")
    assert abs(d.mean()) <= 3e-4 * mean and np.sqrt((d ** 2).mean()) <= 0.05 * mean
    assert np.median(blocks) <= 1e-4 and np.quantile(blocks, .99) <= 0.12
