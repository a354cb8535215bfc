"""TEST INFRASTRUCTURE: generate tests/golden/*.npz from the REFERENCE's own code (oracle/_ref, built
from /root/reference by oracle/Makefile).  Run here, in the container that has /root/reference; the
fixtures (inputs + expected outputs only, no reference source) are committed so the GPU box and
later rounds can check against them without the reference tree.

    python oracle/gen_golden.py

Each case stores: scene name, W, H, spp, integrator, the seed block used (kernel layout), and the
reference's colors[(H-1-y)*W+x], pixels[y*W+x], seeds after the last pass (kernel layout).
The reference's CPU loop reads seeds row-flipped (main-cpu.c:63-66), so the kernel-layout block is
flipped before it is handed to the reference and flipped back afterwards.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(REPO, "path-tracer_amd"))
import oracle_binding as ob  # noqa: E402
import pyptgpu as pt  # noqa: E402

CASES = [  # scene, W, H, spp, integrator
    ("cornell-large", 4, 3, 1, 0),
    ("simple", 32, 24, 1, 0), ("simple", 32, 24, 4, 0), ("simple", 32, 24, 4, 1),
    ("cornell-large", 32, 24, 1, 0), ("cornell-large", 32, 24, 4, 0), ("cornell-large", 32, 24, 4, 1),
    ("caustic", 32, 24, 1, 0), ("caustic", 32, 24, 4, 0), ("caustic", 32, 24, 4, 1),
    ("materials", 32, 24, 1, 0), ("materials", 32, 24, 4, 0), ("materials", 32, 24, 4, 1),
    ("obj-model", 32, 24, 1, 0), ("obj-model", 32, 24, 4, 0), ("obj-model", 32, 24, 2, 1),
    ("cornell", 32, 24, 1, 0), ("cornell", 32, 24, 4, 0), ("cornell", 32, 24, 2, 1),
    ("complex", 32, 24, 2, 0), ("table-and-mirror", 24, 16, 2, 0), ("stacked-spheres", 32, 24, 2, 0),
    ("simple", 37, 19, 3, 0),          # ragged: not a multiple of the 8x8 wave tile
    ("cornell-original", 32, 24, 4, 0),
    ("obj-model-bunny", 32, 24, 2, 0), ("obj-model-bunny", 32, 24, 2, 1),   # 5 050 triangles (data/scenes)
]


def reference_render(name, w, h, spp, integrator):
    sc = pt.LoadedScene(name, w, h)
    seeds_kernel = pt.seed_block(w, h)
    r = ob.Render(sc, pt.flip_rows(seeds_kernel, w, h), ob.SEEDS_CPU_LAYOUT, integrator)
    r.pass_ref(spp)
    return seeds_kernel, r.colors, r.pixels, pt.flip_rows(r.seeds, w, h)


def main():
    out = os.path.join(REPO, "tests", "golden")
    os.makedirs(out, exist_ok=True)
    for name, w, h, spp, integ in CASES:
        seeds, colors, pixels, seeds_after = reference_render(name, w, h, spp, integ)
        fn = os.path.join(out, f"{name}_{w}x{h}x{spp}_{'pt' if integ == 0 else 'dl'}.npz")
        np.savez_compressed(fn, scene=name, width=w, height=h, spp=spp, integrator=integ,
                            colors=colors, pixels=pixels, seeds_after=seeds_after)
        print(fn, "nan px", int(np.isnan(colors).any(1).sum()))
    # RNG known answers straight from the reference's GetRandom
    import ctypes as C
    kat = []
    for s in [(2, 2), (12345, 67890), (1804289383, 846930886), (0xffffffff, 0xffffffff), (65535, 65536)]:
        a, b = C.c_uint32(s[0]), C.c_uint32(s[1])
        row = []
        for _ in range(8):
            f = ob.ref().ref_get_random(C.byref(a), C.byref(b))
            row.append((np.float32(f).view(np.uint32).item(), a.value, b.value))
        kat.append((s, row))
    np.savez_compressed(os.path.join(out, "rng_kat.npz"),
                        start=np.array([k[0] for k in kat], dtype=np.uint32),
                        draws=np.array([k[1] for k in kat], dtype=np.uint32))
    # scene-reader digests: counts + CRC of the record bytes as loaded here (geometry only fields)
    import zlib
    digests = {}
    for scn in sorted(os.listdir(os.path.join(REPO, "data", "scenes"))):
        n = scn[:-4]
        sc = pt.LoadedScene(n, 64, 48)
        digests[n] = (sc.object_count, sc.light_count, int(sc.scene.material_count),
                      zlib.crc32(sc.objects.tobytes()), zlib.crc32(sc.materials.tobytes()))
    np.savez_compressed(os.path.join(out, "scene_digests.npz"),
                        names=np.array(list(digests.keys())),
                        values=np.array(list(digests.values()), dtype=np.uint64))
    print("wrote", len(CASES), "cases")


if __name__ == "__main__":
    main()
