"""The CPU oracle (oracle/liboracle.so) against: the reference's recorded known answers
(SURVEY.md A.2), the committed golden vectors (generated from the reference's own code by
oracle/gen_golden.py), and - where /root/reference is present - the reference's geomfunc.h itself,
function by function.  Everything here is bit-exact (L0 in SURVEY.md 8-c)."""
import ctypes as C
import os

import numpy as np
import pytest

import oracle_binding as ob
import pyptgpu as pt
from conftest import GOLDEN, golden_cases, load_golden, same_bits


def _u32(f):
    return np.float32(f).view(np.uint32).item()


def test_rng_known_answers_from_survey():
    # seeds (2,2): r=0.644916534 -> (73938, 36000); 0.521057129 -> (310613539, 648000000) ...
    a, b = C.c_uint32(2), C.c_uint32(2)
    want = [(0.644916534, 73938, 36000), (0.521057129, 310613539, 648000000),
            (0.513873935, 1420908254, 820233887), (0.599504828, 822138303, 915258515)]
    for r, s0, s1 in want:
        f = ob.oracle().oracle_get_random(C.byref(a), C.byref(b))
        assert abs(f - r) < 5e-9 and (a.value, b.value) == (s0, s1)
    a, b = C.c_uint32(12345), C.c_uint32(67890)
    d = [ob.oracle().oracle_get_random(C.byref(a), C.byref(b)) for _ in range(2)]
    assert abs(d[0] - 0.808948636) < 5e-9 and abs(d[1] - 0.669275999) < 5e-9


def test_rng_golden_vectors():
    z = np.load(os.path.join(GOLDEN, "rng_kat.npz"), allow_pickle=False)
    for start, draws in zip(z["start"], z["draws"]):
        a, b = C.c_uint32(int(start[0])), C.c_uint32(int(start[1]))
        for bits, s0, s1 in draws:
            f = ob.oracle().oracle_get_random(C.byref(a), C.byref(b))
            assert (_u32(f), a.value, b.value) == (int(bits), int(s0), int(s1))
            assert 0.0 <= f < 1.0


def test_known_answer_cornell_large_4x3():
    # SURVEY.md A.2, recorded from the reference's main-cpu.c: CPU seed layout, default rand() seeds
    sc = pt.LoadedScene("cornell-large", 4, 3)
    r = ob.Render(sc, pt.seed_block(4, 3), ob.SEEDS_CPU_LAYOUT).pass_oracle(1)
    hx = lambda v: [float(x).hex() for x in v]
    assert hx(r.colors[0]) == ['0x1.4b91fa0000000p-4', '0x1.b1344a0000000p-6', '0x1.d4c2fa0000000p-6']
    assert hx(r.colors[8]) == ['0x1.defbe00000000p-4', '0x1.3d62d20000000p-5', '0x1.930d560000000p-5']
    assert hx(r.colors[11]) == ['0x1.3a23b80000000p-2', '0x1.9fd28e0000000p-4', '0x1.9fd28e0000000p-4']
    assert ' '.join('%08x' % p for p in r.pixels) == \
        "00413a60 00261622 00484545 005a5a95 0004040c 006360a2 00b1b0b0 00adaaaa 00333151 001f0d16 009e9bdb 008d8d8d"
    assert list(r.seeds[:4]) == [733597813, 219210262, 1451188610, 883924873]


def test_known_answer_simple_config1():
    # scenes/simple.txt 640x480x16 (config 1): mean radiance, pixels[0], seeds[0]
    sc = pt.LoadedScene("simple", 64, 48)
    r = ob.Render(sc, pt.seed_block(64, 48), ob.SEEDS_CPU_LAYOUT).pass_oracle(2)
    assert np.allclose(r.colors.mean(0), (0.705441, 0.741627, 0.77822), atol=2e-6)
    assert r.pixels[0] == 0x00aaa19b and r.seeds[0] == 178938625
    sc = pt.LoadedScene("simple", 640, 480)
    r = ob.Render(sc, pt.seed_block(640, 480), ob.SEEDS_CPU_LAYOUT).pass_oracle(16, threads=8)
    assert np.allclose(r.colors.astype(np.float64).mean(0), (0.708543, 0.744298, 0.781497), atol=2e-6)
    assert r.pixels[0] == 0x00a9a19a and r.seeds[0] == 1170838333


@pytest.mark.parametrize("fn", golden_cases())
def test_oracle_reproduces_golden(fn):
    g = load_golden(fn)
    sc = pt.LoadedScene(g["scene"], g["w"], g["h"])
    r = ob.Render(sc, pt.seed_block(g["w"], g["h"]), ob.SEEDS_KERNEL_LAYOUT, g["integrator"])
    r.pass_oracle(g["spp"])
    assert same_bits(r.colors, g["colors"])         # NaN positions included
    assert (r.pixels == g["pixels"]).all()
    assert (r.seeds == g["seeds_after"]).all()


def test_threads_and_counting_do_not_change_results():
    sc = pt.LoadedScene("materials", 40, 30)
    seeds = pt.seed_block(40, 30)
    a = ob.Render(sc, seeds).pass_oracle(3, threads=1)
    b = ob.Render(sc, seeds).pass_oracle(3, threads=4)
    ctr = ob.Counters()
    c = ob.Render(sc, seeds).pass_oracle_counted(ctr, 3)
    for r in (b, c):
        assert same_bits(a.colors, r.colors) and (a.pixels == r.pixels).all() and (a.seeds == r.seeds).all()
    assert ctr.paths == 40 * 30 * 3 and ctr.extension_rays >= ctr.paths
    assert ctr.rng_draws >= 2 * ctr.paths


def test_flops_per_path_match_survey():
    # F = 20 S + 52 T + 70 LS + 120 R; per-path values probed on the reference
    for name, w, h, spp, want in [("simple", 160, 120, 4, 652), ("caustic", 128, 96, 4, 246),
                                  ("cornell-large", 128, 96, 4, 3113), ("cornell", 32, 24, 2, 1655154)]:
        sc = pt.LoadedScene(name, w, h)
        ctr = ob.Counters()
        ob.Render(sc, pt.seed_block(w, h)).pass_oracle_counted(ctr, spp)
        got = ctr.flops() / ctr.paths
        assert abs(got - want) / want < 0.08, (name, got, want)


def test_against_reference_functions(have_ref):
    """Function-level bit equality with the reference's geomfunc.h on random inputs."""
    if not have_ref:
        pytest.skip("oracle/_ref not built")
    O, R = ob.oracle(), ob.ref()
    rng = np.random.default_rng(7)
    for name in ("cornell", "materials", "obj-model", "simple"):
        sc = pt.LoadedScene(name, 32, 24)
        objs, n, L = sc.objects.ctypes.data_as(C.c_void_p), sc.object_count, sc.light_count
        for _ in range(300):
            o = rng.uniform(-60, 60, 3) + (0, 45, 100)
            d = rng.normal(size=3)
            d /= np.linalg.norm(d)
            ray = pt.Ray(pt.Vec3(*o), pt.Vec3(*d))
            t1, t2, i1, i2 = C.c_float(), C.c_float(), C.c_uint32(7), C.c_uint32(7)
            h1 = O.oracle_intersect(objs, n, C.byref(ray), C.byref(t1), C.byref(i1))
            h2 = R.ref_intersect(objs, n, C.byref(ray), C.byref(t2), C.byref(i2))
            assert (h1, _u32(t1.value), i1.value) == (h2, _u32(t2.value), i2.value)
            mt = float(rng.uniform(1, 300))
            assert O.oracle_intersect_p(objs, n, C.byref(ray), mt) == R.ref_intersect_p(objs, n, C.byref(ray), mt)
            # light sampling from a point with a random normal: same radiance bits, same RNG state
            s = rng.integers(2, 2**31, 4, dtype=np.uint32)
            a0, a1, b0, b1 = (C.c_uint32(int(x)) for x in (s[0], s[1], s[0], s[1]))
            hp, nn = pt.Vec3(*(o * 0.5)), pt.Vec3(*d)
            r1, r2 = pt.Vec3(), pt.Vec3()
            O.oracle_sample_lights(objs, n, L, C.byref(a0), C.byref(a1), C.byref(hp), C.byref(nn), C.byref(r1))
            R.ref_sample_lights(objs, n, L, C.byref(b0), C.byref(b1), C.byref(hp), C.byref(nn), C.byref(r2))
            assert bytes(r1) == bytes(r2) and (a0.value, a1.value) == (b0.value, b1.value)
            for integ in (0, 1):
                a0, a1, b0, b1 = (C.c_uint32(int(x)) for x in (s[2], s[3], s[2], s[3]))
                O.oracle_radiance(objs, n, L, C.byref(ray), C.byref(a0), C.byref(a1), C.byref(sc.scene),
                                  sc.materials.ctypes.data_as(C.c_void_p), integ, C.byref(r1))
                R.ref_radiance(objs, n, L, C.byref(ray), C.byref(b0), C.byref(b1), C.byref(sc.scene),
                               sc.materials.ctypes.data_as(C.c_void_p), integ, C.byref(r2))
                assert bytes(r1) == bytes(r2) and (a0.value, a1.value) == (b0.value, b1.value)


def test_against_reference_full_pass_with_camera_views(have_ref):
    if not have_ref:
        pytest.skip("oracle/_ref not built")
    for name, yaw, pitch in [("cornell-large", 0.3, -0.1), ("materials", 5.9, 0.2), ("caustic", 0.0, 0.4)]:
        sc = pt.LoadedScene(name, 24, 16)
        sc.set_view(yaw, pitch)
        seeds = pt.seed_block(24, 16)
        for integ in (0, 1):
            a = ob.Render(sc, seeds, ob.SEEDS_CPU_LAYOUT, integ).pass_oracle(3)
            b = ob.Render(sc, seeds, ob.SEEDS_CPU_LAYOUT, integ).pass_ref(3)
            assert same_bits(a.colors, b.colors) and (a.pixels == b.pixels).all() and (a.seeds == b.seeds).all()


def test_nan_black_firefly_packs_like_the_reference():
    # quirk A.3-5: a NaN radiance poisons the running mean and packs to 0x80000000
    c = pt.Vec3(0.5, 0.5, 0.5)
    s = pt.Vec3(float("nan"), float("nan"), float("nan"))
    px = ob.oracle().oracle_accumulate(C.byref(c), C.byref(s), 3, 2.2)
    assert px == 0x80000000 and np.isnan(c.x)
    c = pt.Vec3(0, 0, 0)
    s = pt.Vec3(2.0, 0.25, -1.0)
    px = ob.oracle().oracle_accumulate(C.byref(c), C.byref(s), 0, 1.0)
    assert px == (255 | (64 << 8) | (0 << 16))
