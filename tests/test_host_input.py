"""The reference's interactive input, headless (libpthost pthost_input_*): displayfunc.c:408-447 updateCamera,
:450-458 update, :482-493 reshape, :496-622 keyboard / keyboardUp, :628-659 motion / mouse, :661-671 shift;
main-gpu.c:885-913 reInitViewpointDependentBuffers.

displayfunc.c cannot be built here (GL/freeglut.h), so the event arithmetic is PARITY UNPINNED against the reference's
binary; it is checked against a second, independent restatement written here from the reference's expressions with
every promotion explicit (numpy float32 / float64), and the camera basis every event ends in is checked against the
reference's own updateCameraBasis (oracle/_ref, camera.c) bit for bit."""
import ctypes as C

import numpy as np

import oracle_binding as ob
import pyptgpu as pt

f32, f64 = np.float32, np.float64
TWO_PI, PI_OVER_TWO = f32(6.28318530717958647693), f32(1.57079632679489661923)


def vnorm(v):   # vec.h:20: float l = 1.f / sqrt(vdot(v, v)) - sqrt and division in double; vsmul in float
    dot = f32(f32(f32(v[0] * v[0]) + f32(v[1] * v[1])) + f32(v[2] * v[2]))
    l = f32(f64(1.0) / np.sqrt(f64(dot)))
    return [f32(l * v[0]), f32(l * v[1]), f32(l * v[2])]


class Model:
    """updateCamera / motion restated with explicit promotions (the basis is left to _ref)."""

    def __init__(self, cam):
        self.orig = [f32(cam.orig.x), f32(cam.orig.y), f32(cam.orig.z)]
        self.target = [f32(cam.target.x), f32(cam.target.y), f32(cam.target.z)]
        self.yaw, self.pitch = f32(cam.yaw), f32(cam.pitch)
        self.keys, self.shift, self.mouse, self.button, self.time_before = set(), False, (0, 0), 0, 0
        self.original = (list(self.orig), list(self.target), self.yaw, self.pitch)

    def idle(self, now, direction, lateral):
        delta = now - (1 if self.time_before else 0)            # `timeBefore || 0`: a logical or upstream
        self.time_before = now
        speed = f32(f32(f32(delta) * f32(0.0001)) * f32(10 if self.shift else 1))
        move, moved = [f32(0), f32(0), f32(0)], False
        if self.keys & set("wWsS"):
            k = f32(1 if self.keys & set("wW") else -1)
            move = [f32(k * f32(d)) for d in direction]
            moved = True
        if self.keys & set("aAdD"):
            k = f32(1 if self.keys & set("dD") else -1)
            side = [f32(k * c) for c in vnorm([f32(c) for c in lateral])]
            move = [f32(side[i] + move[i]) for i in range(3)]
            moved = True
        if moved:
            move = [f32(speed * c) for c in vnorm(move)]
            self.orig = [f32(self.orig[i] + move[i]) for i in range(3)]
            self.target = [f32(self.target[i] + move[i]) for i in range(3)]
        return moved

    def motion(self, x, y):
        dx, dy = self.mouse[0] - x, self.mouse[1] - y
        if dx == 0 and dy == 0:
            return False
        if self.button == 0:
            self.yaw = f32(f64(self.yaw) + f64(dx) * f64(0.01))
            self.yaw = f32(f64(self.yaw) - f64(TWO_PI) * np.floor(f64(f32(self.yaw / TWO_PI))))
            self.pitch = f32(f64(self.pitch) + f64(-dy) * f64(0.01))
            self.pitch = -PI_OVER_TWO if self.pitch < -PI_OVER_TWO else (PI_OVER_TWO if self.pitch > PI_OVER_TWO else self.pitch)
        self.mouse = (x, y)
        return True


def check(inp, model, have_ref):
    cam = inp.camera
    assert [f32(cam.orig.x), f32(cam.orig.y), f32(cam.orig.z)] == model.orig
    assert [f32(cam.target.x), f32(cam.target.y), f32(cam.target.z)] == model.target
    assert f32(cam.yaw) == model.yaw and f32(cam.pitch) == model.pitch
    if have_ref:                                                 # camera.c:8-32 on the same orig / yaw / pitch / size
        ref = pt.Camera.from_buffer_copy(cam)
        ref.dir = ref.x = ref.y = pt.Vec3(0, 0, 0)
        ob.ref().ref_update_camera_basis(C.byref(ref))
        assert bytes(ref) == bytes(cam)


def test_random_sessions_against_the_explicit_restatement_and_the_reference_basis(have_ref):
    rng = np.random.default_rng(11)
    for scene, w, h in (("cornell", 640, 480), ("obj-model", 800, 600), ("simple", 320, 200)):
        sc = pt.LoadedScene(scene, w, h)
        inp, model = pt.HostInput(sc.camera, w, h), Model(sc.camera)
        check(inp, model, have_ref)
        now = 0
        for _ in range(400):
            ev = int(rng.integers(0, 8))
            if ev == 0:
                key = "wasdWASDqx"[int(rng.integers(0, 10))]
                assert inp.key(key) == pt.EV_NONE
                if key in "wasdWASD":
                    model.keys |= {key.lower(), key.upper()}
            elif ev == 1:
                key = "wasdWASD"[int(rng.integers(0, 8))]
                assert inp.key_up(key) == pt.EV_NONE
                model.keys -= {key.lower(), key.upper()}
            elif ev == 2:
                model.shift = bool(rng.integers(0, 2))
                assert inp.shift(model.shift) == pt.EV_NONE
            elif ev == 3:
                b, x, y = int(rng.integers(0, 3)), int(rng.integers(0, w)), int(rng.integers(0, h))
                model.button, model.mouse = b, (x, y)
                assert inp.mouse(b, x, y) == pt.EV_NONE          # the motion() it ends in sees no movement
            elif ev in (4, 5):
                x, y = int(rng.integers(0, w)), int(rng.integers(0, h))
                moved = model.motion(x, y)
                assert inp.motion(x, y) == (pt.EV_CAMERA if moved else pt.EV_NONE)
            else:
                now += int(rng.integers(1, 40))
                cam = inp.camera
                moved = model.idle(now, cam.dir.tuple(), cam.x.tuple())
                assert inp.idle(now) == (pt.EV_CAMERA if moved else pt.EV_NONE)
            check(inp, model, have_ref)
        # ' ' goes back to the camera of the scene file (displayfunc.c:521-525), held keys stay held
        assert inp.key(" ") == pt.EV_CAMERA
        model.orig, model.target, model.yaw, model.pitch = list(model.original[0]), list(model.original[1]), model.original[2], model.original[3]
        check(inp, model, have_ref)


def test_known_answers_of_single_events():
    sc = pt.LoadedScene("cornell", 640, 480)                     # camera 0 45 205.6, looking down -z
    inp = pt.HostInput(sc.camera, 640, 480)
    # the step is delta * 1e-4 with delta = the time since GLUT started (not since the last frame): w held, idle at
    # 1000 ms then 1016 ms moves 0.1 and then 0.1015 (1016 - 1), not 0.0016
    inp.key("w")
    assert inp.idle(1000) == pt.EV_CAMERA
    assert abs(inp.camera.orig.z - (205.6 - 0.1)) < 1e-4
    assert inp.idle(1016) == pt.EV_CAMERA
    assert abs(inp.camera.orig.z - (205.6 - 0.1 - 0.1015)) < 1e-4 and inp.camera.orig.x == 0
    inp.key_up("w")
    assert inp.idle(1032) == pt.EV_NONE
    # d: along +x of the film; with shift ten times as far
    inp.key("d"); inp.shift(True)
    z = inp.camera.orig.z
    assert inp.idle(2000) == pt.EV_CAMERA
    assert abs(inp.camera.orig.x - 1.999) < 1e-3 and inp.camera.orig.z == z
    inp.key_up("D"); inp.shift(False)                            # either case releases both
    assert inp.idle(2100) == pt.EV_NONE
    # mouse: press at (100, 100), drag 50 px left and 20 px up the screen (y grows downwards): yaw += 0.5,
    # pitch -= 0.2 (0.01 rad per pixel; deltaY = mouseY - y, pitch += -deltaY * 0.01)
    assert inp.mouse(0, 100, 100) == pt.EV_NONE
    assert inp.motion(50, 80) == pt.EV_CAMERA
    assert abs(inp.camera.yaw - 0.5) < 1e-6 and abs(inp.camera.pitch + 0.2) < 1e-6
    assert inp.motion(50, 80) == pt.EV_NONE
    inp.motion(50, 1000)                                          # pitch clamps at +pi/2
    assert inp.camera.pitch == float(PI_OVER_TWO)
    inp.motion(50 + 700, 1000)                                    # yaw wraps into [0, 2 pi)
    assert 0 <= inp.camera.yaw < float(TWO_PI) and abs(inp.camera.yaw - (0.5 - 7.0 + 4 * np.pi)) < 1e-5
    # another button still resets the samples (displayfunc.c:646), without turning the camera
    yaw = inp.camera.yaw
    inp.mouse(2, 10, 10)
    assert inp.motion(20, 20) == pt.EV_CAMERA and inp.camera.yaw == yaw
    # the other keys
    assert inp.key("p") == pt.EV_SNAPSHOT and inp.key("P") == pt.EV_SNAPSHOT
    assert inp.key(27) == pt.EV_QUIT
    assert inp.key(" ") == pt.EV_CAMERA
    assert inp.camera.yaw == 0 and inp.camera.pitch == 0 and inp.camera.orig.tuple() == sc.camera.orig.tuple()


def test_reshape_reallocates_with_the_next_seed_block_and_keeps_the_first_aspect_ratio():
    w, h = 64, 48
    sc = pt.LoadedScene("cornell", w, h)
    inp = pt.HostInput(sc.camera, w, h)
    # the first block went to setUpOpenCL's allocation; GLUT's first reshape (window creation) draws the second
    assert inp.reshape(w, h) == (pt.EV_CAMERA | pt.EV_REALLOC)
    assert (inp.next_seeds() == pt.seed_block(w, h, rank=1)).all()
    assert inp.reshape(32, 20) == (pt.EV_CAMERA | pt.EV_REALLOC)
    seeds = inp.next_seeds()
    everything = np.empty(2 * (2 * w * h) + 2 * 32 * 20, dtype=np.uint32)
    pt.host().pthost_seed_block(everything.ctypes.data_as(C.c_void_p), len(everything), 0)
    assert len(seeds) == 2 * 32 * 20 and (seeds == everything[2 * 2 * w * h:]).all()
    # camera.width / height are the command line's, untouched by reshape (displayfunc.c:132-133, 482-493): the film keeps
    # the first window's aspect ratio
    cam = inp.camera
    assert (cam.width, cam.height) == (w, h) and abs(cam.x.x / cam.y.y - w / h) < 1e-6
