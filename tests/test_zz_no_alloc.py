"""ptgpu_render_static (include/ptgpu.h): the entry point that allocates nothing.

It renders into image buffers the CALLER owns, with the wavefront kernel's work queue in a static
device array of the library instead of the buffer ptgpu_resize allocates.  It has no counterpart in
the reference; the bar is the entry point it is a variant of: the same bits - colours, packed pixels,
RNG state - as ptgpu_render (wavefront kernel, work queue on) from the same inputs.  Two calls in a
row on one context, the second on a SMALLER frame: the static array keeps the first call's per-tile
words, which the second call must not trip over.  Both math builds.

The caller's buffers come from hipMalloc on the HIP runtime the library is linked to (pyptgpu.DeviceBuffer:
device memory the library never sees being made); every render call goes through the C-ABI.
Collected last on purpose (the round's end runs pytest -x).
"""
import numpy as np
import pytest

import pyptgpu as pt

pytestmark = pytest.mark.gpu


def reference_render(sc, w, h, seeds, math, first, n):
    """ptgpu_resize + ptgpu_render with the options ptgpu_render_static is defined against."""
    g = pt.Gpu(0, pt.INTEGRATOR_PT, kernel=pt.KERNEL_WAVEFRONT, math=math, work_queue=1)
    g.set_scene(sc)
    g.resize(w, h, seeds)
    g.sample = 0
    if first:
        g.render(first)
    g.render(n)
    g.finish()
    out = (g.colors(), g.pixels(), g.seeds())
    g.close()
    return out


def static_render(g, w, h, seeds, first, n):
    """The caller's side of ptgpu_render_static: own device buffers, raw addresses, no resize."""
    colors = pt.DeviceBuffer(w * h * pt.COLOR_STRIDE_FLOATS * 4)
    pixels = pt.DeviceBuffer(w * h * 4)
    seeds_d = pt.DeviceBuffer(2 * w * h * 4)
    try:
        seeds_d.upload(seeds)
        done = 0
        for chunk in ([first] if first else []) + [n]:   # `first` passes, then n more: the running mean is read back in
            g.render_static(w, h, done, chunk, colors.ptr, pixels.ptr, seeds_d.ptr)
            done += chunk
        g.finish()
        c = colors.download(np.float32).reshape(-1, pt.COLOR_STRIDE_FLOATS)
        assert (c[:, 3] == 0).all()
        return (np.ascontiguousarray(c[:, :3]), pixels.download(np.uint32), seeds_d.download(np.uint32))
    finally:
        for b in (colors, pixels, seeds_d):
            b.free()


@pytest.mark.parametrize("math", [pt.MATH_STRICT, pt.MATH_FAST], ids=["strict", "fast"])
def test_render_static_equals_render_twice_in_a_row_second_frame_smaller(math):
    g = pt.Gpu(0, pt.INTEGRATOR_PT, math=math)   # options left at AUTO: the entry point picks its kernel itself
    try:
        for scene, w, h, first, n in (("cornell", 200, 136, 2, 3), ("cornell", 72, 40, 0, 4), ("materials", 37, 29, 1, 2),
                                          ("obj-model", 96, 64, 1, 2)):   # (a hierarchy scene: cooperative walks)
            sc = pt.LoadedScene(scene, w, h)
            seeds = pt.seed_block(w, h)
            g.set_scene(sc)
            got = static_render(g, w, h, seeds, first, n)
            want = reference_render(sc, w, h, seeds, math, first, n)
            assert (got[2] == want[2]).all(), (scene, w, h, "seeds")
            assert got[0].tobytes() == want[0].tobytes(), (scene, w, h, "colours")
            assert (got[1] == want[1]).all(), (scene, w, h, "pixels")
    finally:
        g.close()


def test_render_static_refuses_what_it_cannot_hold_and_needs_no_resize():
    g = pt.Gpu(0, pt.INTEGRATOR_PT)
    buf = pt.DeviceBuffer(4096)
    try:
        sc = pt.LoadedScene("simple", 16, 16)
        g.set_scene(sc)
        with pytest.raises(pt.GpuError, match="beyond"):   # 8192 x 8192 = 1 048 576 tiles > 262 142
            g.render_static(8192, 8192, 0, 1, buf.ptr, buf.ptr, buf.ptr)
        with pytest.raises(pt.GpuError, match="null"):
            g.render_static(16, 16, 0, 1, None, buf.ptr, buf.ptr)
        with pytest.raises(pt.GpuError, match="aligned"):
            g.render_static(16, 16, 0, 1, buf.ptr.value + 4, buf.ptr, buf.ptr)
        # the context is still usable, and never saw ptgpu_resize
        seeds = pt.seed_block(16, 16)
        got = static_render(g, 16, 16, seeds, 0, 2)
        want = reference_render(sc, 16, 16, seeds, pt.MATH_FAST, 0, 2)
        assert got[0].tobytes() == want[0].tobytes() and (got[1] == want[1]).all() and (got[2] == want[2]).all()
    finally:
        g.close()
        buf.free()


def test_render_static_allocates_no_device_memory_and_one_context_at_a_time_owns_the_queue():
    """(1) hipMemGetInfo's free bytes are the same before and after a run of calls (after one warm-up call: the first
    call on a device resolves the array's address and may load code).  (2) The static array is one per GPU: while a
    context's call may still be in flight (no ptgpu_finish yet) another context's call on that GPU is refused with
    PTGPU_ERR_STATE instead of racing; after the owner's ptgpu_finish the other context renders, and its image equals
    ptgpu_render's."""
    w, h = 96, 64
    sc = pt.LoadedScene("cornell", w, h)
    seeds = pt.seed_block(w, h)
    a, b = pt.Gpu(0, pt.INTEGRATOR_PT), pt.Gpu(0, pt.INTEGRATOR_PT)
    bufs = [pt.DeviceBuffer(w * h * pt.COLOR_STRIDE_FLOATS * 4), pt.DeviceBuffer(w * h * 4), pt.DeviceBuffer(2 * w * h * 4)]
    try:
        a.set_scene(sc)
        b.set_scene(sc)
        bufs[2].upload(seeds)
        a.render_static(w, h, 0, 1, bufs[0].ptr, bufs[1].ptr, bufs[2].ptr)   # warm-up
        a.finish()
        free_before = pt.device_memory_free()
        for i, (ww, hh) in enumerate(((96, 64), (40, 24), (96, 64))):
            a.render_static(ww, hh, 1 + i, 2, bufs[0].ptr, bufs[1].ptr, bufs[2].ptr)
        a.finish()
        assert pt.device_memory_free() == free_before
        # ownership
        a.render_static(w, h, 0, 4, bufs[0].ptr, bufs[1].ptr, bufs[2].ptr)
        with pytest.raises(pt.GpuError, match="in use by another context") as e:
            b.render_static(w, h, 0, 1, bufs[0].ptr, bufs[1].ptr, bufs[2].ptr)
        assert e.value.code == pt.ERR_STATE
        a.finish()
        bufs[2].upload(seeds)
        got = static_render(b, w, h, seeds, 0, 2)
        want = reference_render(sc, w, h, seeds, pt.MATH_FAST, 0, 2)
        assert got[0].tobytes() == want[0].tobytes() and (got[1] == want[1]).all() and (got[2] == want[2]).all()
        b.finish()
        a.render_static(w, h, 0, 1, bufs[0].ptr, bufs[1].ptr, bufs[2].ptr)   # and back
        a.finish()
    finally:
        a.close()
        b.close()
        for x in bufs:
            x.free()
