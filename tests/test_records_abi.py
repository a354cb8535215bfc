"""Record layouts and C-ABI surface (CPU only, no compute calls on the GPU library)."""
import ctypes as C
import os
import re

import pytest

import oracle_binding as ob
import pyptgpu as pt
from conftest import REPO


def test_record_sizes_match_reference_layout():
    # 12 / 24 / 84 / 44 / 76 / 32 bytes
    assert [C.sizeof(t) for t in (pt.Vec3, pt.Ray, pt.Object, pt.Material, pt.Camera, pt.Scene)] == \
        [12, 24, 84, 44, 76, 32]
    assert pt.Object.emission.offset == 4 and pt.Object.color.offset == 16
    assert pt.Object.radius.offset == 28 and pt.Object.center.offset == 32
    assert pt.Object.p1.offset == 44 and pt.Object.p2.offset == 56 and pt.Object.p3.offset == 68
    assert pt.Object.area.offset == 80
    assert pt.Material.kd.offset == 4 and pt.Material.ks.offset == 16 and pt.Material.sigma.offset == 28
    assert pt.Material.A.offset == 36 and pt.Material.B.offset == 40
    assert pt.Camera.dir.offset == 40 and pt.Camera.x.offset == 52 and pt.Camera.y.offset == 64
    assert pt.Scene.material_count.offset == 28


def test_record_sizes_against_reference_build(have_ref):
    if not have_ref:
        pytest.skip("oracle/_ref not built (no /root/reference here)")
    out = (C.c_int * 6)()
    ob.ref().ref_record_sizes(out)
    assert list(out) == [12, 24, 84, 44, 76, 32]


def _declared(header, prefix):
    text = open(os.path.join(REPO, "include", header)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(%s_\w+)\s*\(" % prefix, text)))


def test_libptgpu_exports_every_declared_symbol():
    names = _declared("ptgpu.h", "ptgpu")
    assert len(names) >= 18
    lib = C.CDLL(os.path.join(REPO, "path-tracer_amd", "libptgpu.so"))
    for n in names:
        assert hasattr(lib, n), n
    assert set(names) == set(pt.GPU_SYMBOLS)


def test_libpthost_exports_every_declared_symbol():
    names = _declared("pthost.h", "pthost")
    lib = C.CDLL(os.path.join(REPO, "path-tracer_amd", "libpthost.so"))
    for n in names:
        assert hasattr(lib, n), n


def test_product_does_not_link_the_oracle():
    # the radiance path must not route through the CPU oracle in any form
    for lib in ("libptgpu.so", "libpthost.so"):
        blob = open(os.path.join(REPO, "path-tracer_amd", lib), "rb").read()
        assert b"liboracle" not in blob and b"oracle_render_pass" not in blob and b"libptref" not in blob
    for root, _, files in os.walk(os.path.join(REPO, "path-tracer_amd")):
        for f in files:
            if f.endswith((".py", ".c", ".h", ".hip", ".inc", ".cpp")):
                src = open(os.path.join(root, f)).read()
                assert "oracle_binding" not in src and "liboracle" not in src, os.path.join(root, f)


def test_gpu_library_fails_loudly_without_device():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(pt.GpuError):
        pt.Gpu(0)
