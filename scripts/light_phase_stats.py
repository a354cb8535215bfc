"""Event counts of the wavefront light phase on cornell 1024x768 (measurement build ablate3 of scripts/ablate.sh)."""
import ctypes as C, os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "path-tracer_amd"))
os.environ["PTGPU_LIB"] = os.path.join(REPO, "path-tracer_amd", "variants", "libptgpu_ablate3.so")
import pyptgpu as pt
scn = sys.argv[1] if len(sys.argv) > 1 else "cornell"
w, h, n = 1024, 768, 2
sc = pt.LoadedScene(scn, w, h)
g = pt.Gpu(0, 0, kernel=pt.KERNEL_WAVEFRONT); g.set_scene(sc); g.resize(w, h, pt.seed_block(w, h))
buf = (C.c_ulonglong * 16)()
lib = pt.gpu()
lib.ptgpu_debug_stats(buf, 1)
g.render(n); g.finish()
lib.ptgpu_debug_stats(buf, 0)
names = ["light iterations", "skipped (whole wave behind)", "lanes that may face", "lanes that want a ray", "rays to near queue",
         "far batches", "rays in far batches", "near batches", "rays in near batches", "light-phase calls", "diffuse lanes at calls",
         "sample batches", "candidates in sample batches", "sphere groups scanned by near batches",
         "triangle groups scanned by near batches", "lights spanned by near batches"]
v = [int(x) for x in buf]
paths = w * h * n
for k, name in enumerate(names):
    print(f"{name:32s} {v[k]:>14d}   per path {v[k] / paths:10.3f}")
it = v[0]
print(f"skipped iterations {v[1] / it:.3f}; in the other iterations: candidates {v[2] / (it - v[1]):.1f} / 64, of which want a ray {v[3] / max(v[2], 1):.3f}")
print(f"sample batch fill {v[12] / max(v[11], 1):.1f}")
print(f"near share of rays {v[4] / max(v[3], 1):.3f}; far batch fill {v[6] / max(v[5], 1):.1f}; near batch fill {v[8] / max(v[7], 1):.1f}; diffuse lanes per call {v[10] / max(v[9], 1):.1f}")
print(f"per near batch: {v[13] / max(v[7], 1):.2f} spheres, {v[14] / max(v[7], 1):.2f} triangle groups, {v[15] / max(v[7], 1):.1f} lights spanned")
