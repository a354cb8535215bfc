"""How much do the shadow tests against the wall spheres matter in the REFERENCE's own arithmetic?
CPU only: the oracle renders cornell.txt with the full scan, and with the six r = 1e4 spheres taken out
of every light's candidate set (mathematically they cannot occlude a segment between two points inside
them; numerically the float discriminant can report the sphere a ray starts on)."""
import os, sys
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(REPO, "path-tracer_amd")); sys.path.insert(0, os.path.join(REPO, "oracle"))
import pyptgpu as pt, oracle_binding as ob
name = sys.argv[1] if len(sys.argv) > 1 else "cornell"
w, h, spp = 128, 96, 8
sc = pt.LoadedScene(name, w, h)
seeds = pt.seed_block(w, h)
full = ob.Render(sc, seeds, ob.SEEDS_KERNEL_LAYOUT, 0).pass_oracle(spp, threads=8)
keep = np.ones((sc.light_count, sc.object_count), dtype=np.uint8)
o = sc.objects
big = (o["type"] == 0) & (o["radius"] >= 1000)
def xyz(v): return np.asarray(v, dtype=np.float64)
emit = ((o["emission"][:, 0] != 0) | (o["emission"][:, 2] != 0)) & (o["type"] != 2)   # upstream looks at x twice (A.3-1)
lights = np.nonzero(emit)[0][: sc.light_count]
mode = sys.argv[2] if len(sys.argv) > 2 else "inside"
for li, l in enumerate(lights):
    for s_ in np.nonzero(big)[0]:
        c, r = xyz(o["center"][s_]), float(o["radius"][s_])
        if mode == "all":
            keep[li, s_] = 0
        elif o["type"][l] == 1:
            v = np.stack([xyz(o["p1"][l]), xyz(o["p2"][l]), xyz(o["p3"][l])])
            if (np.linalg.norm(v - c, axis=1) < r * (1 - 1e-5)).all():
                keep[li, s_] = 0
print("light x big-sphere pairs kept", int(keep[:, big].sum()), "of", len(lights) * int(big.sum()))
with ob.shadow_candidates(keep):
    cut = ob.Render(sc, seeds, ob.SEEDS_KERNEL_LAYOUT, 0).pass_oracle(spp, threads=8)
ok = ~(np.isnan(full.colors).any(1) | np.isnan(cut.colors).any(1))
d = cut.colors[ok].astype(np.float64) - full.colors[ok]
print(f"{name} {w}x{h}x{spp}: spheres dropped {int(big.sum())}; seeds equal {(full.seeds == cut.seeds).all()}")
print(f"pixels that differ {(np.abs(d).max(1) > 0).mean():.4f}; mean radiance full {full.colors[ok].mean():.5f} cut {cut.colors[ok].mean():.5f} "
      f"(relative {d.mean() / full.colors[ok].mean():+.2e}); rmse {np.sqrt((d ** 2).mean()):.3g}; max {np.abs(d).max():.3g}")
