"""ctypes bindings over the two C-ABI libraries of this repo (plumbing for tests and bench.py).

  libpthost.so  plain-C host side: scene reader, OBJ ingestion, camera basis, seeds, PPM
                (include/pthost.h)
  libptgpu.so   the HIP radiance path (include/ptgpu.h)

Nothing here computes anything: every call goes straight through the C-ABI.  There is no CPU
fallback; `gpu()` raises if libptgpu.so is missing.
"""
import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
DATA_DIR = os.path.join(REPO, "data")


class Vec3(C.Structure):
    _fields_ = [("x", C.c_float), ("y", C.c_float), ("z", C.c_float)]

    def tuple(self):
        return (self.x, self.y, self.z)


class Ray(C.Structure):
    _fields_ = [("o", Vec3), ("d", Vec3)]


class Object(C.Structure):
    _fields_ = [("type", C.c_int16), ("material_id", C.c_int16), ("emission", Vec3), ("color", Vec3),
                ("radius", C.c_float), ("center", Vec3), ("p1", Vec3), ("p2", Vec3), ("p3", Vec3),
                ("area", C.c_float)]


class Material(C.Structure):
    _fields_ = [("type", C.c_int16), ("pad_", C.c_int16), ("kd", Vec3), ("ks", Vec3),
                ("sigma", C.c_float), ("roughness", C.c_float), ("A", C.c_float), ("B", C.c_float)]


class Camera(C.Structure):
    _fields_ = [("orig", Vec3), ("target", Vec3), ("pitch", C.c_float), ("yaw", C.c_float),
                ("width", C.c_float), ("height", C.c_float), ("dir", Vec3), ("x", Vec3), ("y", Vec3)]


class Scene(C.Structure):
    _fields_ = [("gamma", C.c_float), ("sky1", Vec3), ("sky2", Vec3), ("material_count", C.c_uint32)]


class HostScene(C.Structure):
    _fields_ = [("scene", Scene), ("camera", Camera), ("objects", C.POINTER(Object)),
                ("object_count", C.c_uint32), ("light_count", C.c_uint32),
                ("materials", C.POINTER(Material)), ("title", C.c_char * 100)]


class Input(C.Structure):
    """pthost_input (include/pthost.h): the reference's interactive state, headless."""
    _fields_ = [("camera", Camera), ("original_camera", Camera), ("key_states", C.c_ubyte * 256),
                ("shift_pressed", C.c_int), ("mouse_button", C.c_int), ("mouse_x", C.c_int), ("mouse_y", C.c_int),
                ("time_before", C.c_int), ("width", C.c_int), ("height", C.c_int), ("rand_consumed", C.c_ulonglong)]


EV_NONE, EV_CAMERA, EV_REALLOC, EV_SNAPSHOT, EV_QUIT = 0, 1, 2, 4, 8

assert C.sizeof(Vec3) == 12 and C.sizeof(Ray) == 24 and C.sizeof(Object) == 84
assert C.sizeof(Material) == 44 and C.sizeof(Camera) == 76 and C.sizeof(Scene) == 32

OBJECT_DTYPE = np.dtype([("type", "<i2"), ("material_id", "<i2"), ("emission", "<f4", 3),
                         ("color", "<f4", 3), ("radius", "<f4"), ("center", "<f4", 3),
                         ("p1", "<f4", 3), ("p2", "<f4", 3), ("p3", "<f4", 3), ("area", "<f4")])
MATERIAL_DTYPE = np.dtype([("type", "<i2"), ("pad_", "<i2"), ("kd", "<f4", 3), ("ks", "<f4", 3),
                           ("sigma", "<f4"), ("roughness", "<f4"), ("A", "<f4"), ("B", "<f4")])
assert OBJECT_DTYPE.itemsize == 84 and MATERIAL_DTYPE.itemsize == 44

_host = None
_gpu = None


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


def host():
    """libpthost.so, loaded once."""
    global _host
    if _host is None:
        lib = C.CDLL(os.path.join(HERE, "libpthost.so"))
        lib.pthost_last_error.restype = C.c_char_p
        lib.pthost_load_scene.argtypes = [C.c_char_p, C.c_char_p, C.c_int, C.c_int, C.POINTER(HostScene)]
        lib.pthost_free_scene.argtypes = [C.POINTER(HostScene)]
        lib.pthost_update_camera_basis.argtypes = [C.POINTER(Camera)]
        lib.pthost_seed_block.argtypes = [C.c_void_p, C.c_size_t, C.c_uint64]
        lib.pthost_seed_block.restype = None
        lib.pthost_flip_rows_u32x2.argtypes = [C.c_void_p, C.c_int, C.c_int]
        lib.pthost_flip_rows_u32x2.restype = None
        lib.pthost_write_ppm.argtypes = [C.c_char_p, C.c_void_p, C.c_int, C.c_int]
        lib.pthost_obj_load.argtypes = [C.c_char_p, C.POINTER(C.POINTER(C.c_float)), C.POINTER(C.c_uint32),
                                        C.POINTER(C.POINTER(C.c_int32)), C.POINTER(C.c_uint32),
                                        C.POINTER(C.c_uint32)]
        lib.pthost_obj_parse.argtypes = [C.c_char_p, C.c_size_t, C.POINTER(C.POINTER(C.c_float)),
                                         C.POINTER(C.c_uint32), C.POINTER(C.POINTER(C.c_int32)),
                                         C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
        lib.pthost_snapshot_name.argtypes = [C.c_char_p, C.c_size_t, C.c_char_p, C.c_int, C.c_double]
        lib.pthost_snapshot_name.restype = None
        lib.pthost_captions.argtypes = [C.c_char_p, C.c_char_p, C.c_size_t, C.c_double, C.c_double,
                                        C.c_int, C.c_int, C.c_int, C.c_int]
        lib.pthost_captions.restype = None
        lib.pthost_checkpoint_write.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        lib.pthost_checkpoint_read.argtypes = [C.c_char_p, C.POINTER(C.c_int), C.POINTER(C.c_int),
                                               C.POINTER(C.c_int), C.POINTER(C.POINTER(C.c_float)),
                                               C.POINTER(C.POINTER(C.c_uint32))]
        lib.pthost_frame_time_budget.argtypes = [C.c_int]
        lib.pthost_frame_time_budget.restype = C.c_float
        ip = C.POINTER(Input)
        lib.pthost_input_init.argtypes = [ip, C.POINTER(Camera), C.c_int, C.c_int]
        lib.pthost_input_init.restype = None
        lib.pthost_input_idle.argtypes = [ip, C.c_int]
        lib.pthost_input_keyboard.argtypes = [ip, C.c_ubyte]
        lib.pthost_input_keyboard_up.argtypes = [ip, C.c_ubyte]
        lib.pthost_input_shift.argtypes = [ip, C.c_int]
        lib.pthost_input_mouse.argtypes = [ip, C.c_int, C.c_int, C.c_int]
        lib.pthost_input_motion.argtypes = [ip, C.c_int, C.c_int]
        lib.pthost_input_reshape.argtypes = [ip, C.c_int, C.c_int]
        lib.pthost_input_next_seeds.argtypes = [ip, C.c_void_p]
        lib.pthost_input_next_seeds.restype = None
        _host = lib
    return _host


class HostError(RuntimeError):
    pass


class LoadedScene:
    """A scene read by libpthost, with numpy views over the malloc()ed record arrays."""

    def __init__(self, name_or_path, width, height, model_dir=DATA_DIR):
        path = name_or_path
        if not os.path.exists(path):
            path = os.path.join(DATA_DIR, "scenes", name_or_path + ".txt")
        self.raw = HostScene()
        rc = host().pthost_load_scene(path.encode(), model_dir.encode() if model_dir else None,
                                      width, height, C.byref(self.raw))
        if rc != 0:
            raise HostError(host().pthost_last_error().decode())
        host().pthost_update_camera_basis(C.byref(self.raw.camera))
        self.width, self.height = width, height
        n = self.raw.object_count
        m = self.raw.scene.material_count
        self.objects = np.ctypeslib.as_array(
            C.cast(self.raw.objects, C.POINTER(C.c_uint8)), shape=(n * 84,)).view(OBJECT_DTYPE).copy()
        self.materials = np.ctypeslib.as_array(
            C.cast(self.raw.materials, C.POINTER(C.c_uint8)), shape=(m * 44,)).view(MATERIAL_DTYPE).copy()
        self.camera = Camera.from_buffer_copy(self.raw.camera)
        self.scene = Scene.from_buffer_copy(self.raw.scene)
        self.object_count = int(n)
        self.light_count = int(self.raw.light_count)
        self.title = self.raw.title.decode()
        host().pthost_free_scene(C.byref(self.raw))

    def set_view(self, yaw=0.0, pitch=0.0, orig=None):
        self.camera.yaw, self.camera.pitch = yaw, pitch
        if orig is not None:
            self.camera.orig = Vec3(*orig)
        host().pthost_update_camera_basis(C.byref(self.camera))


class HostInput:
    """The reference's GLUT callbacks as calls (pthost_input_*): each returns the EV_* mask of what the caller must do."""

    def __init__(self, camera, width, height):
        self.state = Input()
        host().pthost_input_init(C.byref(self.state), C.byref(camera), width, height)

    @property
    def camera(self):
        return Camera.from_buffer_copy(self.state.camera)

    def idle(self, time_ms):
        return host().pthost_input_idle(C.byref(self.state), time_ms)

    def key(self, key):
        return host().pthost_input_keyboard(C.byref(self.state), ord(key) if isinstance(key, str) else key)

    def key_up(self, key):
        return host().pthost_input_keyboard_up(C.byref(self.state), ord(key) if isinstance(key, str) else key)

    def shift(self, pressed):
        return host().pthost_input_shift(C.byref(self.state), 1 if pressed else 0)

    def mouse(self, button, x, y):
        return host().pthost_input_mouse(C.byref(self.state), button, x, y)

    def motion(self, x, y):
        return host().pthost_input_motion(C.byref(self.state), x, y)

    def reshape(self, width, height):
        return host().pthost_input_reshape(C.byref(self.state), width, height)

    def next_seeds(self):
        seeds = np.empty(2 * self.state.width * self.state.height, dtype=np.uint32)
        host().pthost_input_next_seeds(C.byref(self.state), _ptr(seeds))
        return seeds


def seed_block(width, height, rank=0):
    """2*W*H rand()-compatible seeds; rank g takes the g-th consecutive block."""
    n = 2 * width * height
    seeds = np.empty(n, dtype=np.uint32)
    host().pthost_seed_block(_ptr(seeds), n, rank * n)
    return seeds


def flip_rows(seeds, width, height):
    out = np.ascontiguousarray(seeds).copy()
    host().pthost_flip_rows_u32x2(_ptr(out), width, height)
    return out


def _take_obj(v, nv, idx, nc, nf):
    verts = (np.ctypeslib.as_array(v, shape=(nv.value * 3,)).copy() if nv.value and v
             else np.zeros(0, np.float32))
    corners = (np.ctypeslib.as_array(idx, shape=(nc.value,)).copy() if nc.value and idx
               else np.zeros(0, np.int32))
    libc = C.CDLL(None)
    libc.free.argtypes = [C.c_void_p]
    libc.free(v)
    libc.free(idx)
    return verts.reshape(-1, 3), corners, nf.value


def obj_load(path):
    v = C.POINTER(C.c_float)()
    idx = C.POINTER(C.c_int32)()
    nv, nc, nf = C.c_uint32(), C.c_uint32(), C.c_uint32()
    rc = host().pthost_obj_load(path.encode(), C.byref(v), C.byref(nv), C.byref(idx), C.byref(nc), C.byref(nf))
    if rc != 0:
        raise HostError(host().pthost_last_error().decode())
    return _take_obj(v, nv, idx, nc, nf)


def checkpoint_write(path, width, height, sample, colors, seeds):
    colors = np.ascontiguousarray(colors, dtype=np.float32)
    seeds = np.ascontiguousarray(seeds, dtype=np.uint32)
    if host().pthost_checkpoint_write(path.encode(), width, height, sample, _ptr(colors), _ptr(seeds)) != 0:
        raise HostError(host().pthost_last_error().decode())


def checkpoint_read(path):
    w, h, k = C.c_int(), C.c_int(), C.c_int()
    c = C.POINTER(C.c_float)()
    s = C.POINTER(C.c_uint32)()
    if host().pthost_checkpoint_read(path.encode(), C.byref(w), C.byref(h), C.byref(k), C.byref(c), C.byref(s)) != 0:
        raise HostError(host().pthost_last_error().decode())
    n = w.value * h.value
    colors = np.ctypeslib.as_array(c, shape=(n, 3)).copy()
    seeds = np.ctypeslib.as_array(s, shape=(2 * n,)).copy()
    libc = C.CDLL(None)
    libc.free.argtypes = [C.c_void_p]
    libc.free(c)
    libc.free(s)
    return w.value, h.value, k.value, colors, seeds


def obj_parse(data: bytes):
    v = C.POINTER(C.c_float)()
    idx = C.POINTER(C.c_int32)()
    nv, nc, nf = C.c_uint32(), C.c_uint32(), C.c_uint32()
    rc = host().pthost_obj_parse(data, len(data), C.byref(v), C.byref(nv), C.byref(idx), C.byref(nc), C.byref(nf))
    if rc != 0:
        raise HostError(host().pthost_last_error().decode())
    return _take_obj(v, nv, idx, nc, nf)


# ---------------------------------------------------------------------------------------------
# libptgpu.so (HIP).  No fallback: a missing library or a missing GPU raises.
# ---------------------------------------------------------------------------------------------
OPT_KERNEL, OPT_MATH, OPT_LDS, OPT_STREAMS, OPT_PASSES_PER_LAUNCH, OPT_WAVES_PER_GROUP = 0, 1, 2, 3, 4, 5
OPT_SHADOW_CULL = 6
OPT_RAY_BINS = 7
OPT_BVH = 8
OPT_WORK_QUEUE = 9
BVH_OFF, BVH_ON, BVH_AUTO = 0, 1, 2
CULL_OFF, CULL_SAFE, CULL_TIGHT, CULL_AUTO, CULL_ENCLOSED, CULL_BALL, CULL_ENCLOSED_SELF = 0, 1, 2, 3, 4, 5, 6
KERNEL_MEGA, KERNEL_WAVEFRONT, KERNEL_AUTO = 0, 1, 2
MATH_FAST, MATH_STRICT = 0, 1
INTEGRATOR_PT, INTEGRATOR_DL = 0, 1
COLOR_STRIDE_FLOATS = 4  # PTGPU_COLOR_STRIDE_FLOATS: the device sample buffer is one float4 per pixel


class SceneInfo(C.Structure):
    _fields_ = ([(n, C.c_uint32) for n in ("primitives", "spheres", "triangles", "lights", "segments",
                                          "lds_bytes", "shadow_groups")]
                + [("shadow_tests_safe", C.c_float), ("shadow_tests_tight", C.c_float),
                   ("shadow_tests_enclosed", C.c_float), ("clusters", C.c_uint32), ("clustered_triangles", C.c_uint32),
                   ("bvh_nodes", C.c_uint32), ("bvh_primitives", C.c_uint32), ("convex_clusters", C.c_uint32)])


GPU_SYMBOLS = ["ptgpu_create", "ptgpu_destroy", "ptgpu_set_option", "ptgpu_set_scene", "ptgpu_set_camera",
               "ptgpu_resize", "ptgpu_render_pass", "ptgpu_render", "ptgpu_render_timed", "ptgpu_finish",
               "ptgpu_read_pixels", "ptgpu_read_colors", "ptgpu_read_seeds", "ptgpu_write_colors",
               "ptgpu_colors_device_ptr", "ptgpu_stream", "ptgpu_scale_and_repack", "ptgpu_get_scene_info",
               "ptgpu_shadow_candidates", "ptgpu_cluster_info", "ptgpu_last_error", "ptgpu_render_static",
               "ptgpu_get_launch_info", "ptgpu_work_counters", "ptgpu_probe_rays"]


class LaunchInfo(C.Structure):
    _fields_ = [("kernel", C.c_int), ("math", C.c_int), ("instance", C.c_int), ("work_queue", C.c_int),
                ("unit_passes", C.c_int), ("passes_per_launch", C.c_int), ("regions", C.c_int),
                ("lds_bytes", C.c_uint), ("bvh_nodes_only", C.c_int), ("shadow_cull", C.c_int), ("ray_bins", C.c_int)]


INSTANCE_NAMES = {0: "plain", 1: "bvh", 2: "hybrid", 3: "ball"}
WORK_SLOTS = ("sphere_tests", "triangle_tests_mt", "triangle_tests_planes", "bvh_nodes", "light_samples",
              "light_points", "plane_side_tests", "segment_near_tests", "ball_point_tests", "extension_rays",
              "shadow_rays", "rng_draws", "paths", "node_lane_slots", "prim_lane_slots", "convex_plane_tests")


ERR_ARGUMENT, ERR_STATE, ERR_DEVICE, ERR_MEMORY = -1, -2, -3, -4   # ptgpu.h ptgpu_status


class GpuError(RuntimeError):
    code = None   # the ptgpu_status a C-ABI call returned, where one did


VARIANT_DIR = os.path.join(HERE, "variants")
_variants = {}


def variant_path(name):
    """A measurement / test build of the library (scripts/build_variants.sh): count, giveup."""
    return os.path.join(VARIANT_DIR, "libptgpu_%s.so" % name)


def hip_runtimes_mapped():
    """Paths of every libamdhip64 mapped into this process (/proc/self/maps)."""
    seen = []
    with open("/proc/self/maps") as f:
        for line in f:
            if "libamdhip64" in line:
                path = line.split()[-1]
                if path not in seen:
                    seen.append(path)
    return seen


def check_single_hip_runtime():
    """libptgpu.so needs libamdhip64.so.7 and takes whichever copy of that SONAME is already mapped; torch loads its
    bundled copy by path.  Loaded in the order torch -> libptgpu there is one runtime; in the order libptgpu -> torch
    there are two, and the one that comes second does not find the GPU (or worse, both run it).  Refuse that state."""
    mapped = hip_runtimes_mapped()
    if len(mapped) > 1:
        raise GpuError("two HIP runtimes are mapped into this process (%s): import torch (or whatever else brings its own "
                       "libamdhip64) BEFORE the first pyptgpu.gpu() / ptgpu_create, so that libptgpu.so binds to that copy"
                       % ", ".join(mapped))


def gpu(path=None):
    """libptgpu.so (or, with `path`, a variant build of it), loaded once per path.  Raises if the HIP
    extension has not been built, or if loading it leaves two HIP runtimes in the process."""
    global _gpu
    check_single_hip_runtime()
    if path is not None:
        if path not in _variants:
            if not os.path.exists(path):
                raise GpuError("%s is missing: build it with scripts/build_variants.sh" % path)
            _variants[path] = _bind(C.CDLL(path))
        return _variants[path]
    if _gpu is None:
        path = os.environ.get("PTGPU_LIB") or os.path.join(HERE, "libptgpu.so")  # PTGPU_LIB: A/B builds
        if not os.path.exists(path):
            raise GpuError("libptgpu.so is missing: build it with `make -C path-tracer_amd/csrc` "
                           "(there is no CPU fallback for the radiance path)")
        _gpu = _bind(C.CDLL(path))
        check_single_hip_runtime()
    return _gpu


def _bind(lib):
    lib.ptgpu_last_error.restype = C.c_char_p
    vp = C.c_void_p
    lib.ptgpu_create.argtypes = [C.POINTER(vp), C.c_int, C.c_int]
    lib.ptgpu_destroy.argtypes = [vp]
    lib.ptgpu_set_option.argtypes = [vp, C.c_int, C.c_int]
    lib.ptgpu_set_scene.argtypes = [vp, vp, C.c_uint32, C.c_uint32, vp, C.c_uint32, C.POINTER(Scene)]
    lib.ptgpu_set_camera.argtypes = [vp, C.POINTER(Camera)]
    lib.ptgpu_resize.argtypes = [vp, C.c_int, C.c_int, vp]
    lib.ptgpu_render_pass.argtypes = [vp, C.c_int]
    lib.ptgpu_render.argtypes = [vp, C.c_int, C.c_int]
    lib.ptgpu_render_timed.argtypes = [vp, C.c_int, C.c_int, C.POINTER(C.c_float)]
    lib.ptgpu_finish.argtypes = [vp]
    lib.ptgpu_render_static.argtypes = [vp, C.c_int, C.c_int, C.c_int, C.c_int, vp, vp, vp]
    lib.ptgpu_read_pixels.argtypes = [vp, vp]
    lib.ptgpu_read_colors.argtypes = [vp, vp]
    lib.ptgpu_read_seeds.argtypes = [vp, vp]
    lib.ptgpu_write_colors.argtypes = [vp, vp]
    lib.ptgpu_colors_device_ptr.argtypes = [vp, C.POINTER(vp)]
    lib.ptgpu_stream.argtypes = [vp, C.POINTER(vp)]
    lib.ptgpu_scale_and_repack.argtypes = [vp, C.c_float]
    lib.ptgpu_get_scene_info.argtypes = [vp, C.POINTER(SceneInfo)]
    lib.ptgpu_shadow_candidates.argtypes = [vp, C.c_uint32, C.c_uint32, C.c_int, vp, C.POINTER(C.c_uint32)]
    if hasattr(lib, "ptgpu_cluster_info"):   # (absent from A/B builds of older revisions, scripts/build_base.sh)
        lib.ptgpu_cluster_info.argtypes = [vp, C.c_uint32, C.c_uint32, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
    lib.ptgpu_get_launch_info.argtypes = [vp, C.POINTER(LaunchInfo)]
    lib.ptgpu_work_counters.argtypes = [vp, C.POINTER(C.c_ulonglong), C.c_int]
    lib.ptgpu_probe_rays.argtypes = [vp, C.c_int] + [C.c_void_p] * 6 + [C.c_int]
    return lib


_hip = None


def hip_runtime():
    """The HIP runtime libptgpu.so itself is linked to (found among this process's mappings), for
    callers that own device buffers (ptgpu_render_static).  A second copy of the runtime in one
    process - torch ships its own - does not find the GPU once this one has it."""
    global _hip
    if _hip is None:
        gpu()
        path = None
        with open("/proc/self/maps") as f:
            for line in f:
                if "libamdhip64" in line:
                    path = line.split()[-1]
                    break
        if path is None:
            raise GpuError("libamdhip64 is not mapped: libptgpu.so did not load its HIP runtime")
        lib = C.CDLL(path)
        lib.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
        lib.hipFree.argtypes = [C.c_void_p]
        lib.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
        lib.hipMemset.argtypes = [C.c_void_p, C.c_int, C.c_size_t]
        lib.hipGetErrorString.restype = C.c_char_p
        lib.hipGetErrorString.argtypes = [C.c_int]
        lib.hipMemGetInfo.argtypes = [C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]
        _hip = lib
    return _hip


def device_memory_free():
    """hipMemGetInfo's free bytes on the current device (tests: an entry point that must not allocate)."""
    free, total = C.c_size_t(), C.c_size_t()
    e = hip_runtime().hipMemGetInfo(C.byref(free), C.byref(total))
    if e != 0:
        raise GpuError("hipMemGetInfo: HIP error %d" % e)
    return free.value


class DeviceBuffer:
    """Device memory the CALLER owns (hipMalloc through hip_runtime())."""

    def __init__(self, nbytes):
        self.lib = hip_runtime()
        self.nbytes = nbytes
        self.ptr = C.c_void_p()
        self._ok(self.lib.hipMalloc(C.byref(self.ptr), max(nbytes, 1)))
        self._ok(self.lib.hipMemset(self.ptr, 0, max(nbytes, 1)))

    def _ok(self, e):
        if e != 0:
            raise GpuError("HIP error %d: %s" % (e, self.lib.hipGetErrorString(e).decode()))

    def upload(self, array):
        a = np.ascontiguousarray(array)
        assert a.nbytes == self.nbytes
        self._ok(self.lib.hipMemcpy(self.ptr, _ptr(a), a.nbytes, 1))   # hipMemcpyHostToDevice

    def download(self, dtype):
        out = np.empty(self.nbytes // np.dtype(dtype).itemsize, dtype=dtype)
        self._ok(self.lib.hipMemcpy(_ptr(out), self.ptr, self.nbytes, 2))   # hipMemcpyDeviceToHost (blocking)
        return out

    def free(self):
        if self.ptr:
            self.lib.hipFree(self.ptr)
            self.ptr = C.c_void_p()

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


def shadow_candidates(sc, mode, light_count=None):
    """keep[l, i]: object i is scanned for a shadow ray towards light l under cull `mode` (host only)."""
    lights = sc.light_count if light_count is None else light_count
    n = C.c_uint32()
    lib = gpu()
    if lib.ptgpu_shadow_candidates(_ptr(sc.objects), sc.object_count, lights, mode, None, C.byref(n)) != 0:
        raise GpuError(lib.ptgpu_last_error().decode())
    keep = np.zeros((n.value, sc.object_count), dtype=np.uint8)
    if lib.ptgpu_shadow_candidates(_ptr(sc.objects), sc.object_count, lights, mode, _ptr(keep), C.byref(n)) != 0:
        raise GpuError(lib.ptgpu_last_error().decode())
    return keep


def cluster_info(sc, light_count=None):
    """(clusters, [face planes of cluster c as a closed convex mesh, 0 = it keeps its triangles] x 8): host only."""
    n, planes = C.c_uint32(), (C.c_uint32 * 8)()
    lib = gpu()
    lights = sc.light_count if light_count is None else light_count
    if lib.ptgpu_cluster_info(_ptr(sc.objects), sc.object_count, lights, C.byref(n), planes) != 0:
        raise GpuError(lib.ptgpu_last_error().decode())
    return n.value, [int(v) for v in planes]


class Gpu:
    """One ptgpu context driven the way main-gpu.c drives its OpenCL objects."""

    def __init__(self, device=0, integrator=INTEGRATOR_PT, kernel=None, math=None, lds=None, streams=None,
                 passes_per_launch=None, waves_per_group=None, shadow_cull=None, ray_bins=None, bvh=None,
                 work_queue=None, lib=None):
        self.lib = gpu(lib)
        self.ctx = C.c_void_p()
        self._ok(self.lib.ptgpu_create(C.byref(self.ctx), device, integrator))
        if kernel is not None:
            self.set_option(OPT_KERNEL, kernel)
        if math is not None:
            self.set_option(OPT_MATH, math)
        if lds is not None:
            self.set_option(OPT_LDS, lds)
        if streams is not None:
            self.set_option(OPT_STREAMS, streams)
        if passes_per_launch is not None:
            self.set_option(OPT_PASSES_PER_LAUNCH, passes_per_launch)
        if waves_per_group is not None:
            self.set_option(OPT_WAVES_PER_GROUP, waves_per_group)
        if shadow_cull is not None:
            self.set_option(OPT_SHADOW_CULL, shadow_cull)
        if ray_bins is not None:
            self.set_option(OPT_RAY_BINS, ray_bins)
        if bvh is not None:
            self.set_option(OPT_BVH, bvh)
        if work_queue is not None:
            self.set_option(OPT_WORK_QUEUE, work_queue)
        self.w = self.h = 0
        self.sample = 0

    def _ok(self, rc):
        if rc != 0:
            e = GpuError("ptgpu status %d: %s" % (rc, self.lib.ptgpu_last_error().decode()))
            e.code = rc
            raise e

    def set_option(self, opt, value):
        self._ok(self.lib.ptgpu_set_option(self.ctx, opt, value))

    def set_scene(self, sc, light_count=None):
        self._ok(self.lib.ptgpu_set_scene(self.ctx, _ptr(sc.objects), sc.object_count,
                                          sc.light_count if light_count is None else light_count,
                                          _ptr(sc.materials), len(sc.materials), C.byref(sc.scene)))
        self._ok(self.lib.ptgpu_set_camera(self.ctx, C.byref(sc.camera)))

    def set_camera(self, camera):
        self._ok(self.lib.ptgpu_set_camera(self.ctx, C.byref(camera)))

    def resize(self, w, h, seeds):
        seeds = np.ascontiguousarray(seeds, dtype=np.uint32)
        if w > 0 and h > 0 and seeds.size != 2 * w * h:
            raise ValueError('seeds must hold 2*W*H words')
        self._ok(self.lib.ptgpu_resize(self.ctx, w, h, _ptr(seeds)))
        self.w, self.h, self.sample = w, h, 0

    def render(self, n_passes=1):
        self._ok(self.lib.ptgpu_render(self.ctx, self.sample, n_passes))
        self.sample += n_passes

    def render_pass(self):
        self._ok(self.lib.ptgpu_render_pass(self.ctx, self.sample))
        self.sample += 1

    def render_timed(self, n_passes):
        ms = C.c_float()
        self._ok(self.lib.ptgpu_render_timed(self.ctx, self.sample, n_passes, C.byref(ms)))
        self.sample += n_passes
        return ms.value

    def finish(self):
        self._ok(self.lib.ptgpu_finish(self.ctx))

    def render_static(self, w, h, first_sample, n_passes, colors_dev, pixels_dev, seeds_dev):
        """ptgpu_render_static: raw device addresses of caller-owned buffers; allocates nothing."""
        self._ok(self.lib.ptgpu_render_static(self.ctx, w, h, first_sample, n_passes, colors_dev, pixels_dev, seeds_dev))

    def pixels(self):
        out = np.empty(self.w * self.h, dtype=np.uint32)
        self._ok(self.lib.ptgpu_read_pixels(self.ctx, _ptr(out)))
        return out

    def colors(self):
        out = np.empty((self.w * self.h, 3), dtype=np.float32)
        self._ok(self.lib.ptgpu_read_colors(self.ctx, _ptr(out)))
        return out

    def seeds(self):
        out = np.empty(2 * self.w * self.h, dtype=np.uint32)
        self._ok(self.lib.ptgpu_read_seeds(self.ctx, _ptr(out)))
        return out

    def write_colors(self, colors):
        colors = np.ascontiguousarray(colors, dtype=np.float32)
        assert colors.size == 3 * self.w * self.h
        self._ok(self.lib.ptgpu_write_colors(self.ctx, _ptr(colors)))

    def colors_device_ptr(self):
        p = C.c_void_p()
        self._ok(self.lib.ptgpu_colors_device_ptr(self.ctx, C.byref(p)))
        return p.value

    def scale_and_repack(self, scale):
        self._ok(self.lib.ptgpu_scale_and_repack(self.ctx, scale))

    def scene_info(self):
        info = SceneInfo()
        self._ok(self.lib.ptgpu_get_scene_info(self.ctx, C.byref(info)))
        return {n: (float if t is C.c_float else int)(getattr(info, n)) for n, t in info._fields_}

    def launch_info(self):
        info = LaunchInfo()
        self._ok(self.lib.ptgpu_get_launch_info(self.ctx, C.byref(info)))
        return {n: int(getattr(info, n)) for n, _ in info._fields_}

    def work_counters(self, reset=True):
        """Counting build only (lib=variant_path("count")): executed work since the last reset."""
        out = (C.c_ulonglong * 16)()
        self._ok(self.lib.ptgpu_work_counters(self.ctx, out, 1 if reset else 0))
        return {n: int(out[i]) for i, n in enumerate(WORK_SLOTS)}

    def probe_rays(self, origins, directions, maxt, cooperative=False):
        """Ray-probe build only (lib=variant_path("probe")): closest hit and occlusion answer of the rays given
        (n x 3, n x 3, n).  Returns t (-1: no hit), rgb of the surface hit (colour + emission), occluded (bool)."""
        o = np.ascontiguousarray(origins, dtype=np.float32).reshape(-1, 3)
        d = np.ascontiguousarray(directions, dtype=np.float32).reshape(-1, 3)
        m = np.ascontiguousarray(maxt, dtype=np.float32).reshape(-1)
        n = len(o)
        assert len(d) == n and len(m) == n
        t = np.empty(n, np.float32); rgb = np.empty((n, 3), np.float32); occ = np.empty(n, np.int32)
        self._ok(self.lib.ptgpu_probe_rays(self.ctx, n, _ptr(o), _ptr(d), _ptr(m), _ptr(t), _ptr(rgb), _ptr(occ),
                                           int(cooperative)))
        return t, rgb, (occ if int(cooperative) == 2 else occ.astype(bool))

    def close(self):
        if self.ctx:
            self.lib.ptgpu_destroy(self.ctx)
            self.ctx = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
