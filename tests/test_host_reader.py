"""libpthost: scene grammar, OBJ ingestion, camera basis, seeds, PPM (CPU only).

Pinned by the reference's recorded behaviour (produced from the reference's own
readScene/main-cpu.c) and, where /root/reference is present, by the reference's vendored tinyobj
and camera.c compiled as they lie (oracle/_ref)."""
import ctypes as C
import os
import zlib

import numpy as np
import pytest

import oracle_binding as ob
import pyptgpu as pt
from conftest import GOLDEN, REPO

# SURVEY.md A.2 "Loaded counts (N, L, M)"
COUNTS = {"simple": (8, 1, 3), "cornell": (143, 123, 3), "cornell-large": (9, 1, 3),
          "materials": (18, 14, 5), "caustic": (3, 1, 3), "obj-model": (330, 2, 3)}
# SURVEY.md A.2 "OBJ faces kept/declared"
FACES = {"cylinder": (123, 124), "teapot": (239, 240), "table": (59, 60), "d20": (19, 20),
         "teapot-800": (1559, 1560), "bunny": (4967, 4968), "cube": (12, 12), "plane-x": (2, 2),
         "plane-y": (2, 2), "plane-z": (2, 2), "plane-y-down": (2, 2), "sponge": (144, 144),
         "light": (60, 60), "monkey": (968, 968)}


@pytest.mark.parametrize("name", sorted(COUNTS))
def test_loaded_counts(name):
    sc = pt.LoadedScene(name, 64, 48)
    assert (sc.object_count, sc.light_count, int(sc.scene.material_count)) == COUNTS[name]


def test_every_scene_loads_and_matches_digest():
    z = np.load(os.path.join(GOLDEN, "scene_digests.npz"), allow_pickle=False)
    for name, v in zip(z["names"], z["values"]):
        sc = pt.LoadedScene(str(name), 64, 48)
        got = (sc.object_count, sc.light_count, int(sc.scene.material_count),
               zlib.crc32(sc.objects.tobytes()), zlib.crc32(sc.materials.tobytes()))
        assert got == tuple(int(x) for x in v), name


def test_scene_details_cornell():
    sc = pt.LoadedScene("cornell", 64, 48)
    t = sc.objects["type"]
    assert (t == 0).sum() == 6 and (t == 1).sum() == 135 and (t == 2).sum() == 2
    assert sc.title == "cornell"
    assert abs(sc.scene.gamma - 2.2) < 1e-6
    # model placeholder is followed by its faces, each a light (quirk A.3-20)
    assert t[6] == 2 and (t[7:7 + 123] == 1).all()
    assert (sc.objects["emission"][7:7 + 123] == 4).all()
    # sphere area 4 pi r^2 in float (displayfunc.c:237)
    r = np.float32(10000)
    assert sc.objects["area"][0] == np.float32(4) * np.float32(np.pi) * r * r


def test_materials_scene_quirks():
    sc = pt.LoadedScene("materials", 64, 48)
    m = sc.materials
    assert list(m["type"]) == [0, 1, 2, 3, 3]  # three built-ins, then user mattes (quirk A.3-19)
    s2 = np.float32(0.65) * np.float32(0.65)
    assert m["A"][3] == np.float32(1) - (s2 / (np.float32(2) * (s2 + np.float32(0.33))))
    assert m["B"][3] == np.float32(0.45) * s2 / (s2 + np.float32(0.09))
    assert sc.objects["material_id"][1] == 3  # "3" selects the first user material
    # default header when the gamma line is absent is covered by cornell-original? (has one) - check defaults
    assert sc.scene.gamma == np.float32(1.0)


def test_missing_header_defaults(tmp_path):
    p = tmp_path / "s.txt"
    p.write_text("camera 0 1 2  3 4 5\nmaterials 0\nobjects 1\nsphere 1  0 0 0  0 5 0  1 1 1  0\n")
    sc = pt.LoadedScene(str(p), 8, 8)
    assert sc.scene.gamma == np.float32(2.2)
    assert sc.scene.sky1.tuple() == (np.float32(0.15), np.float32(0.3), np.float32(0.5))
    assert sc.scene.sky2.tuple() == (1, 1, 1)
    # emission (0, 5, 0) is NOT a light: the upstream zero test looks at x twice (quirk A.3-1)
    assert sc.light_count == 0
    assert sc.camera.target.tuple() == (3, 4, 5) and sc.camera.width == 8


@pytest.mark.parametrize("text,msg", [
    ("gamma 1 sky1 0 0 0 sky2 0 0 0\ncamera 1 2 3\n", "camera"),
    ("camera 0 0 0 0 0 0\nobjects 1\n", "materials"),
    ("camera 0 0 0 0 0 0\nmaterials 0\nobjects 1\nsphere 1 0 0 0\n", "sphere"),
    ("camera 0 0 0 0 0 0\nmaterials 0\nobjects 1\nsphere 1 0 0 0 0 0 0 1 1 1 9\n", "material type"),
    ("camera 0 0 0 0 0 0\nmaterials 1\nmatte 1 1\nobjects 0\n", "matte"),
    ("camera 0 0 0 0 0 0\nmaterials 4294967293\nobjects 0\n", "Too many materials"),   # + 3 built-ins wraps to 0
    ("camera 0 0 0 0 0 0\nmaterials 0\nobjects 16777217\n", "Too many objects"),
    ("camera 0 0 0 0 0 0\nmaterials 0\nobjects 1\nmodel nosuch.obj 0 0 0 1 1 1 0 0 0 1 1 1 0\n", "cannot open"),
])
def test_parse_errors(tmp_path, text, msg):
    p = tmp_path / "bad.txt"
    p.write_text(text)
    with pytest.raises(pt.HostError) as e:
        pt.LoadedScene(str(p), 8, 8)
    assert msg in str(e.value)


def test_missing_file():
    with pytest.raises(pt.HostError):
        pt.LoadedScene("/nonexistent/scene.txt", 8, 8, None)


@pytest.mark.parametrize("name", sorted(FACES))
def test_obj_faces_kept(name):
    path = os.path.join(REPO, "data", "models", name + ".obj")
    verts, corners, nfaces = pt.obj_load(path)
    declared = sum(1 for l in open(path) if l.startswith("f "))
    assert (nfaces, declared) == FACES[name]
    assert len(corners) == 0 or (corners.min() >= 0 and corners.max() < len(verts))


def test_obj_reader_against_vendored_tinyobj(have_ref):
    if not have_ref:
        pytest.skip("oracle/_ref not built")
    mdir = os.path.join(REPO, "data", "models")
    for fn in sorted(os.listdir(mdir)):
        v = C.POINTER(C.c_float)()
        idx = C.POINTER(C.c_int)()
        nv, nc, nf = C.c_uint(), C.c_uint(), C.c_uint()
        rc = ob.ref().ref_tinyobj_load(os.path.join(mdir, fn).encode(), C.byref(v), C.byref(nv),
                                       C.byref(idx), C.byref(nc), C.byref(nf))
        assert rc == 0
        verts, corners, nfaces = pt.obj_load(os.path.join(mdir, fn))
        rv = np.ctypeslib.as_array(v, shape=(nv.value * 3,)).reshape(-1, 3)
        ri = np.ctypeslib.as_array(idx, shape=(nc.value,))
        assert nfaces == nf.value, fn
        assert verts.tobytes() == rv.tobytes(), fn   # bit-identical floats (own decimal parser kept)
        assert (corners == ri).all(), fn


def test_obj_edge_cases():
    tri = b"v 0 0 0\nv 1 0 0\nv 0 1 0\nf 1 2 3\n"
    # single trailing newline: the last line is dropped
    v, c, n = pt.obj_parse(tri)
    assert len(v) == 3 and n == 0
    v, c, n = pt.obj_parse(tri + b"\n")
    assert n == 1 and list(c) == [0, 1, 2]
    # slashes, negative (relative) indices, CRLF, comments, tabs
    v, c, n = pt.obj_parse(b"# c\r\nv 1.5 -2.25 3e2\r\nv\t0 0 0\r\nv 1 1 1\r\nvn 0 0 1\r\nf 1/1/1 2//1 -1\r\n\r\n")
    assert n == 1 and list(c) == [0, 1, 2]
    assert v[0].tolist() == [1.5, -2.25, 300.0]
    # quads stay quads (flags = 0: no triangulation)
    v, c, n = pt.obj_parse(b"v 0 0 0\nv 1 0 0\nv 1 1 0\nv 0 1 0\nf 1 2 3 4\n\n")
    assert n == 1 and list(c) == [0, 1, 2, 3]
    with pytest.raises(pt.HostError):
        pt.obj_parse(b"x")  # no complete line


def test_seed_block_is_glibc_rand():
    libc = C.CDLL(None)
    libc.srand(1)
    want = np.array([libc.rand() for _ in range(5000)], dtype=np.uint32)
    want[want < 2] = 2
    got = np.empty(5000, dtype=np.uint32)
    pt.host().pthost_seed_block(got.ctypes.data_as(C.c_void_p), 5000, 0)
    assert (got == want).all()
    assert list(got[:4]) == [1804289383, 846930886, 1681692777, 1714636915]
    # rank blocks are consecutive and disjoint
    b0, b1 = pt.seed_block(16, 8, 0), pt.seed_block(16, 8, 1)
    assert (np.concatenate([b0, b1]) == want[:512]).all()


def test_flip_rows_roundtrip():
    s = pt.seed_block(5, 3)
    f = pt.flip_rows(s, 5, 3)
    assert (f.reshape(3, 10)[::-1] == s.reshape(3, 10)).all()
    assert (pt.flip_rows(f, 5, 3) == s).all()


def test_camera_basis_known_answer_and_reference(have_ref):
    sc = pt.LoadedScene("cornell", 640, 480)
    cam = sc.camera
    # dir (-0,-0,-1), x (1.0472,0,0), y (0,0.785398,0) at 4:3
    assert cam.dir.tuple() == (-0.0, -0.0, -1.0)
    assert abs(cam.x.x - 1.0472) < 1e-4 and cam.x.y == 0 and cam.x.z == 0
    assert abs(cam.y.y - 0.785398) < 1e-6
    if have_ref:
        rng = np.random.default_rng(1)
        for _ in range(200):
            c1 = pt.Camera.from_buffer_copy(cam)
            c1.yaw, c1.pitch = float(rng.uniform(0, 6.28)), float(rng.uniform(-1.5, 1.5))
            c1.width, c1.height = float(rng.integers(1, 4000)), float(rng.integers(1, 4000))
            c2 = pt.Camera.from_buffer_copy(c1)
            pt.host().pthost_update_camera_basis(C.byref(c1))
            ob.ref().ref_update_camera_basis(C.byref(c2))
            assert bytes(c1) == bytes(c2)


def test_ppm_snapshot_and_names(tmp_path):
    px = np.array([0x00030201, 0x00060504, 0x00090807, 0x000c0b0a], dtype=np.uint32)  # 2x2
    p = tmp_path / "o.ppm"
    assert pt.host().pthost_write_ppm(str(p).encode(), px.ctypes.data_as(C.c_void_p), 2, 2) == 0
    # rows bottom-up, R = low byte (displayfunc.c:508-515)
    assert p.read_text() == "P3\n2 2\n255\n7 8 9 10 11 12 1 2 3 4 5 6 "
    buf = C.create_string_buffer(200)
    pt.host().pthost_snapshot_name(buf, 200, b"cornell", 3152, 501.0)
    assert buf.value == b"cornell-3152-8m21s.ppm"        # the reference's own file name pattern
    pt.host().pthost_snapshot_name(buf, 200, b"skybox", 2413, 49.0)
    assert buf.value == b"skybox-2413-49s.ppm"
    l1, l2 = C.create_string_buffer(256), C.create_string_buffer(256)
    pt.host().pthost_captions(l1, l2, 256, 75.0, 0.5, 40, 4, 100, 50)
    assert l1.value == b"Time:   1m15s  (0.500s frames/s)" and l2.value == b"Passes: 40  (40.0K samples/s)"


def test_frame_budget_schedule():
    f = pt.host().pthost_frame_time_budget
    assert f(0) == 0 and f(19) == 0 and f(20) == 0
    assert abs(f(70) - 0.25) < 1e-7 and abs(f(120) - 0.5) < 1e-7 and abs(f(5000) - 0.5) < 1e-7


def test_checkpoint_roundtrip_and_errors(tmp_path):
    rng = np.random.default_rng(5)
    w, h = 7, 5
    colors = rng.random((w * h, 3), dtype=np.float32)
    colors[3] = np.nan                                   # NaN pixels survive a checkpoint
    seeds = rng.integers(2, 2**32, 2 * w * h, dtype=np.uint32)
    p = str(tmp_path / "c.ckpt")
    pt.checkpoint_write(p, w, h, 41, colors, seeds)
    assert os.path.getsize(p) == 8 + 12 + 12 * w * h + 8 * w * h
    w2, h2, k, c2, s2 = pt.checkpoint_read(p)
    assert (w2, h2, k) == (w, h, 41) and c2.tobytes() == colors.tobytes() and (s2 == seeds).all()
    open(p, "r+b").truncate(100)
    with pytest.raises(pt.HostError, match="truncated"):
        pt.checkpoint_read(p)
    (tmp_path / "x").write_bytes(b"not a checkpoint at all....")
    with pytest.raises(pt.HostError, match="PTCKPT01"):
        pt.checkpoint_read(str(tmp_path / "x"))
    with pytest.raises(pt.HostError):
        pt.checkpoint_read(str(tmp_path / "missing"))
