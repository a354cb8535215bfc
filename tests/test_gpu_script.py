"""ptcli --script: the reference's interactive session without its window.  A script of key, mouse and
reshape events is replayed by the C host on the GPU (strict build) and by this test on the CPU - the same libpthost
event functions driving the oracle pass by pass - and the two final images must be the same bytes; after every event
the camera is the reference's own updateCameraBasis (oracle/_ref) of the same position and angles."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

import oracle_binding as ob
import pyptgpu as pt
from conftest import REPO

pytestmark = pytest.mark.gpu

SCRIPT = """# GLUT's own first reshape (window creation): the interactive run renders with the SECOND seed block
reshape {w} {h}
idle 16 3
key w
idle 500 2
keyup w
mouse 0 100 100
motion 90 95
idle 600 2
key d
shift down
idle 700 1
shift up
keyup d
idle 800 4
key p
reshape {w2} {h2}
idle 900 2
mouse 0 10 10
motion 14 7
idle 1000 3
"""


def replay_on_the_cpu(sc, w, h, text, integrator, have_ref):
    inp = pt.HostInput(sc.camera, w, h)
    render = ob.Render(sc, pt.seed_block(w, h), ob.SEEDS_KERNEL_LAYOUT, integrator)
    snapshot = None
    for line in text.splitlines():
        tok = line.split()
        if not tok or tok[0].startswith("#"):
            continue
        ev, frame = pt.EV_NONE, 0
        if tok[0] == "idle":
            ev, frame = inp.idle(int(tok[1])), int(tok[2])
        elif tok[0] == "key":
            ev = inp.key({"space": " ", "esc": chr(27)}.get(tok[1], tok[1]))
        elif tok[0] == "keyup":
            ev = inp.key_up(tok[1])
        elif tok[0] == "shift":
            ev = inp.shift(tok[1] == "down")
        elif tok[0] == "mouse":
            ev = inp.mouse(int(tok[1]), int(tok[2]), int(tok[3]))
        elif tok[0] == "motion":
            ev = inp.motion(int(tok[1]), int(tok[2]))
        elif tok[0] == "reshape":
            ev = inp.reshape(int(tok[1]), int(tok[2]))
        if have_ref:                                             # camera.c's basis of the same position and angles
            ref = pt.Camera.from_buffer_copy(inp.camera)
            ref.dir = ref.x = ref.y = pt.Vec3(0, 0, 0)
            ob.ref().ref_update_camera_basis(C.byref(ref))
            assert bytes(ref) == bytes(inp.camera), line
        if ev & pt.EV_REALLOC:
            sc.width, sc.height = inp.state.width, inp.state.height
            render = ob.Render(sc, inp.next_seeds(), ob.SEEDS_KERNEL_LAYOUT, integrator)
        if ev & pt.EV_CAMERA:
            sc.camera = inp.camera
            render.sample = 0
        if ev & pt.EV_SNAPSHOT:
            snapshot = (render.pixels.copy(), render.w, render.h, render.sample)
        if frame:
            render.pass_oracle(frame, threads=8)
    return render, snapshot


def read_ppm(path):
    tok = open(path).read().split()
    w, h = int(tok[1]), int(tok[2])
    rgb = np.array(tok[4:], dtype=np.uint32).reshape(h, w, 3)[::-1]   # the file is bottom-up
    return (rgb[..., 0] | (rgb[..., 1] << 8) | (rgb[..., 2] << 16)), w, h


@pytest.mark.parametrize("scene,kernel_file", [("cornell-large", "path-tracing.cl"), ("materials", "ray-tracing.cl")])
def test_scripted_session_matches_the_oracle_driven_by_the_same_events(tmp_path, have_ref, scene, kernel_file):
    w, h, w2, h2 = 48, 36, 40, 24
    text = SCRIPT.format(w=w, h=h, w2=w2, h2=h2)
    script = tmp_path / "session.txt"
    script.write_text(text)
    out = tmp_path / "final.ppm"
    cmd = [os.path.join(REPO, "path-tracer_amd", "ptcli"), "1", "0", kernel_file, str(w), str(h),
           os.path.join(REPO, "data", "scenes", scene + ".txt"), "--math", "strict", "--models", os.path.join(REPO, "data"),
           "--script", str(script), "--out", str(out)]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=180, cwd=tmp_path)
    assert r.returncode == 0, r.stderr
    integrator = 1 if kernel_file.startswith("ray") else 0
    sc = pt.LoadedScene(scene, w, h)
    cpu, snapshot = replay_on_the_cpu(sc, w, h, text, integrator, have_ref)
    # the final image: 3 passes after the last camera change, at the reshaped size, on the third seed block
    assert "%s: 3 passes, %dx%d" % (scene, w2, h2) in r.stdout, r.stdout
    got, gw, gh = read_ppm(out)
    assert (gw, gh) == (w2, h2) and (got == (cpu.pixels.reshape(h2, w2) & 0xffffff)).all()
    # the 'p' snapshot: the frame as displayed when the key came (5 passes since the last camera change)
    shots = [f for f in os.listdir(tmp_path) if f.startswith(scene + "-5-") and f.endswith(".ppm")]
    assert len(shots) == 1, os.listdir(tmp_path)
    got, gw, gh = read_ppm(tmp_path / shots[0])
    assert snapshot[3] == 5 and (gw, gh) == (w, h) and (got == (snapshot[0].reshape(h, w) & 0xffffff)).all()
    # every frame line reports the camera the events left
    assert r.stderr.count("frame ") == 7


def test_script_errors_exit_like_the_reference(tmp_path):
    script = tmp_path / "bad.txt"
    script.write_text("idle 16 1\nwobble 3\n")
    cmd = [os.path.join(REPO, "path-tracer_amd", "ptcli"), "1", "0", "path-tracing.cl", "16", "16",
           os.path.join(REPO, "data", "scenes", "simple.txt"), "--models", os.path.join(REPO, "data"), "--script", str(script)]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=120, cwd=tmp_path)
    assert r.returncode != 0 and "cannot read the event 'wobble'" in r.stderr
    r = subprocess.run(cmd[:-1] + [str(tmp_path / "missing.txt")], capture_output=True, text=True, timeout=120, cwd=tmp_path)
    assert r.returncode != 0 and "Failed to open script file" in r.stderr
