"""TEST INFRASTRUCTURE: ctypes bindings for oracle/liboracle.so (our CPU restatement) and, when it has
been built in this container, oracle/_ref/libptref.so (the reference's own geomfunc.h + camera.c).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this module.
"""
import ctypes as C
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
sys.path.insert(0, os.path.join(REPO, "path-tracer_amd"))
import pyptgpu as pt  # noqa: E402  (record definitions only)


class Counters(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in ("sphere_tests", "triangle_tests", "light_samples",
                                          "extension_rays", "shadow_rays", "rng_draws", "paths")]

    def flops(self):
        """Reference-equivalent flops, SURVEY.md 8-d: F = 20 S + 52 T + 70 LS + 120 R."""
        return (20 * self.sphere_tests + 52 * self.triangle_tests + 70 * self.light_samples
                + 120 * self.extension_rays)

    def as_dict(self):
        return {n: int(getattr(self, n)) for n, _ in self._fields_}


SEEDS_CPU_LAYOUT, SEEDS_KERNEL_LAYOUT = 0, 1
_p = C.c_void_p


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


_oracle = None
_ref = None


def oracle():
    global _oracle
    if _oracle is None:
        lib = C.CDLL(os.path.join(HERE, "liboracle.so"))
        lib.oracle_get_random.restype = C.c_float
        lib.oracle_get_random.argtypes = [C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
        lib.oracle_camera_ray.argtypes = [C.POINTER(pt.Camera), C.c_int, C.c_int, C.c_int, C.c_int,
                                          C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.POINTER(pt.Ray)]
        lib.oracle_intersect.argtypes = [_p, C.c_uint32, C.POINTER(pt.Ray), C.POINTER(C.c_float),
                                         C.POINTER(C.c_uint32)]
        lib.oracle_intersect_p.argtypes = [_p, C.c_uint32, C.POINTER(pt.Ray), C.c_float]
        lib.oracle_sample_lights.argtypes = [_p, C.c_uint32, C.c_uint32, C.POINTER(C.c_uint32),
                                             C.POINTER(C.c_uint32), C.POINTER(pt.Vec3),
                                             C.POINTER(pt.Vec3), C.POINTER(pt.Vec3)]
        lib.oracle_radiance.argtypes = [_p, C.c_uint32, C.c_uint32, C.POINTER(pt.Ray),
                                        C.POINTER(C.c_uint32), C.POINTER(C.c_uint32),
                                        C.POINTER(pt.Scene), _p, C.c_int, C.POINTER(pt.Vec3)]
        lib.oracle_accumulate.restype = C.c_uint32
        lib.oracle_accumulate.argtypes = [C.POINTER(pt.Vec3), C.POINTER(pt.Vec3), C.c_int, C.c_float]
        lib.oracle_render_pass.argtypes = [_p, C.c_uint32, C.c_uint32, _p, C.POINTER(pt.Camera),
                                           C.POINTER(pt.Scene), C.c_int, C.c_int, _p, _p, _p,
                                           C.c_int, C.c_int, C.c_int, C.c_int]
        lib.oracle_render_pass.restype = None
        lib.oracle_render_pass_counted.argtypes = [_p, C.c_uint32, C.c_uint32, _p, C.POINTER(pt.Camera),
                                                   C.POINTER(pt.Scene), C.c_int, C.c_int, _p, _p, _p,
                                                   C.c_int, C.c_int, C.c_int, C.POINTER(Counters)]
        lib.oracle_render_pass_counted.restype = None
        lib.oracle_set_shadow_candidates.argtypes = [_p]
        lib.oracle_set_shadow_candidates.restype = None
        lib.oracle_set_enclosing.argtypes = [_p]
        lib.oracle_set_enclosing.restype = None
        _oracle = lib
    return _oracle


def ref_available():
    return os.path.exists(os.path.join(HERE, "_ref", "libptref.so"))


def ref():
    """The reference's own code (oracle/_ref); only present where /root/reference was available."""
    global _ref
    if _ref is None:
        lib = C.CDLL(os.path.join(HERE, "_ref", "libptref.so"))
        lib.ref_get_random.restype = C.c_float
        lib.ref_get_random.argtypes = [C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
        lib.ref_update_camera_basis.argtypes = [C.POINTER(pt.Camera)]
        lib.ref_intersect.argtypes = [_p, C.c_uint32, C.POINTER(pt.Ray), C.POINTER(C.c_float),
                                      C.POINTER(C.c_uint32)]
        lib.ref_intersect_p.argtypes = [_p, C.c_uint32, C.POINTER(pt.Ray), C.c_float]
        lib.ref_sample_lights.argtypes = [_p, C.c_uint32, C.c_uint32, C.POINTER(C.c_uint32),
                                          C.POINTER(C.c_uint32), C.POINTER(pt.Vec3), C.POINTER(pt.Vec3),
                                          C.POINTER(pt.Vec3)]
        lib.ref_sample_lights.restype = None
        lib.ref_radiance.argtypes = [_p, C.c_uint32, C.c_uint32, C.POINTER(pt.Ray), C.POINTER(C.c_uint32),
                                     C.POINTER(C.c_uint32), C.POINTER(pt.Scene), _p, C.c_int,
                                     C.POINTER(pt.Vec3)]
        lib.ref_radiance.restype = None
        lib.ref_render_pass.argtypes = [_p, C.c_uint32, C.c_uint32, _p, C.POINTER(pt.Camera),
                                        C.POINTER(pt.Scene), C.c_int, C.c_int, _p, _p, _p, C.c_int, C.c_int]
        lib.ref_render_pass.restype = None
        lib.ref_tinyobj_load.argtypes = [C.c_char_p, C.POINTER(C.POINTER(C.c_float)), C.POINTER(C.c_uint),
                                         C.POINTER(C.POINTER(C.c_int)), C.POINTER(C.c_uint),
                                         C.POINTER(C.c_uint)]
        _ref = lib
    return _ref


class shadow_candidates:
    """`with shadow_candidates(keep):` - oracle renders inside replay the product's shadow-ray culling
    (keep = pyptgpu.shadow_candidates(scene, mode)); outside, the reference's full scan."""

    def __init__(self, keep):
        self.keep = np.ascontiguousarray(keep, dtype=np.uint8)
        # the spheres the ENCLOSED family dropped for some light (bytes 2, 3, 4): their far roots are watched
        self.enclosing = np.ascontiguousarray((self.keep >= 2).any(0), dtype=np.uint8)

    def __enter__(self):
        oracle().oracle_set_shadow_candidates(_ptr(self.keep))
        oracle().oracle_set_enclosing(_ptr(self.enclosing) if self.enclosing.any() else None)
        return self

    def __exit__(self, *exc):
        oracle().oracle_set_shadow_candidates(None)
        oracle().oracle_set_enclosing(None)
        return False


class Render:
    """State of a progressive CPU render: colors[(H-1-y)*W+x], pixels[y*W+x], seeds."""

    def __init__(self, scene, seeds, seed_layout=SEEDS_KERNEL_LAYOUT, integrator=0):
        self.scene = scene
        self.w, self.h = scene.width, scene.height
        self.seeds = np.ascontiguousarray(seeds, dtype=np.uint32).copy()
        self.colors = np.zeros((self.w * self.h, 3), dtype=np.float32)
        self.pixels = np.zeros(self.w * self.h, dtype=np.uint32)
        self.sample = 0
        self.layout = seed_layout
        self.integrator = integrator

    def _common(self):
        s = self.scene
        return (_ptr(s.objects), s.object_count, s.light_count, _ptr(s.materials), C.byref(s.camera),
                C.byref(s.scene), self.w, self.h, _ptr(self.seeds), _ptr(self.colors), _ptr(self.pixels))

    def pass_oracle(self, n=1, threads=1):
        for _ in range(n):
            oracle().oracle_render_pass(*self._common(), self.sample, self.integrator, self.layout, threads)
            self.sample += 1
        return self

    def pass_oracle_counted(self, counters, n=1):
        for _ in range(n):
            oracle().oracle_render_pass_counted(*self._common(), self.sample, self.integrator,
                                                self.layout, C.byref(counters))
            self.sample += 1
        return self

    def pass_ref(self, n=1):
        """The reference's code; it only knows the CPU seed layout."""
        assert self.layout == SEEDS_CPU_LAYOUT
        for _ in range(n):
            ref().ref_render_pass(*self._common(), self.sample, self.integrator)
            self.sample += 1
        return self
