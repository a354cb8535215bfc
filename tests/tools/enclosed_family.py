"""The ENCLOSED family of candidate sets on the GPU (round 3): (1) the strict build with each mode equals the CPU
oracle's replay of that mode bit for bit; (2) at 1024x768x4 the strict build with ENCLOSED / ENCLOSED_SELF against
itself with SAFE and TIGHT (pixels that differ, mean); (3) the fast build's bias against strict + SAFE; (4) timing of
the modes and of the library builds named on the command line.
usage: enclosed_family.py [--skip-replay] [lib.so ...]"""
import sys, os, time
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(REPO, "oracle")); sys.path.insert(0, os.path.join(REPO, "path-tracer_amd"))
import pyptgpu as pt
args = sys.argv[1:]
skip_replay = "--skip-replay" in args
libs = [a for a in args if a.endswith(".so")]
MODES = {"safe": pt.CULL_SAFE, "tight": pt.CULL_TIGHT, "enclosed": pt.CULL_ENCLOSED, "self": pt.CULL_ENCLOSED_SELF, "ball": pt.CULL_BALL}
SCENES = ("cornell", "cornell-large", "cornell-original")


def render(sc, w, h, spp, seeds, **kw):
    g = pt.Gpu(0, 0, **kw); g.set_scene(sc); g.resize(w, h, seeds); g.render(spp); g.finish()
    out = (g.colors(), g.seeds(), g.launch_info()); g.close()
    return out


if not skip_replay:
    import oracle_binding as ob
    for name in SCENES:
        w, h, spp = 64, 48, 2
        sc = pt.LoadedScene(name, w, h); seeds = pt.seed_block(w, h)
        for mname in ("enclosed", "self", "ball"):
            with ob.shadow_candidates(pt.shadow_candidates(sc, MODES[mname])):
                cpu = ob.Render(sc, seeds, ob.SEEDS_KERNEL_LAYOUT, 0).pass_oracle(spp, threads=8)
            for kernel in (pt.KERNEL_MEGA, pt.KERNEL_WAVEFRONT):
                for lds in (1, 0):
                    col, sd, li = render(sc, w, h, spp, seeds, math=pt.MATH_STRICT, kernel=kernel, shadow_cull=MODES[mname], bvh=pt.BVH_OFF, lds=lds)
                    same = col.tobytes() == cpu.colors.tobytes() and (sd == cpu.seeds).all()
                    print(f"replay {name:16s} {mname:8s} kernel {kernel} lds {lds}: strict == oracle replay: {same} (cull in effect {li['shadow_cull']})", flush=True)

w, h, spp = 1024, 768, int(os.environ.get("SPP", "4"))
for name in SCENES:
    sc = pt.LoadedScene(name, w, h); seeds = pt.seed_block(w, h)
    strict = {m: render(sc, w, h, spp, seeds, math=pt.MATH_STRICT, shadow_cull=MODES[m], bvh=pt.BVH_OFF)[0] for m in ("safe", "tight", "enclosed", "self")}
    for ref in ("safe", "tight"):
        for m in ("enclosed", "self"):
            a, b = strict[m], strict[ref]
            ok = np.isfinite(a).all(1) & np.isfinite(b).all(1)
            d = a[ok].astype(np.float64) - b[ok]
            print(f"strict {name:16s} {m:8s} vs {ref:5s} {w}x{h}x{spp}: pixels that differ {(np.abs(d).max(1) > 0).sum()} of {ok.sum()}, "
                  f"mean {d.mean() / b[ok].mean():+.2e}, nan px {(~np.isfinite(a).all(1)).sum()} / {(~np.isfinite(b).all(1)).sum()}", flush=True)
    b = strict["safe"]
    for m in ("safe", "tight", "enclosed", "self"):
        a = render(sc, w, h, spp, seeds, math=pt.MATH_FAST, shadow_cull=MODES[m])[0]
        ok = np.isfinite(a).all(1) & np.isfinite(b).all(1)
        d = a[ok].astype(np.float64) - b[ok]; mean = b[ok].astype(np.float64).mean()
        sigma = d.mean(1).std() / np.sqrt(ok.sum()) / mean
        print(f"fast   {name:16s} {m:8s} vs strict safe: bias {d.mean() / mean:+.2e} (sigma {sigma:.1e}), rel rmse {np.sqrt((d ** 2).mean()) / mean:.3e}", flush=True)

# timing: modes with the working tree's library, then the libraries named
for name in SCENES + ("cornell-large@4k",):
    W, H = (3840, 2160) if name.endswith("@4k") else (1024, 768)
    scn = name.split("@")[0]
    sc = pt.LoadedScene(scn, W, H); seeds = pt.seed_block(W, H)
    cases = [(None, m) for m in ("enclosed", "self", "tight")] + [(lib, m) for lib in libs for m in ("enclosed", "self")]
    gs = []
    for lib, m in cases:
        try:
            g = pt.Gpu(0, 0, lib=lib, shadow_cull=MODES[m]); g.set_scene(sc); g.resize(W, H, seeds)
        except pt.GpuError as e:   # (an older build does not know the mode)
            gs.append(None); continue
        info = g.scene_info()
        n = 32 if info["primitives"] * (1 + info["lights"]) > 2000 else (64 if W > 2000 else 128)
        g.render_timed(n); gs.append((g, n))
    best = [1e30] * len(cases)
    for _ in range(3):
        for i, e in enumerate(gs):
            if e: best[i] = min(best[i], e[0].render_timed(e[1]) / e[1])
    for i, (lib, m) in enumerate(cases):
        if gs[i]:
            print(f"time   {name:18s} {os.path.basename(lib) if lib else 'libptgpu.so':24s} {m:8s} {best[i]:8.4f} ms/pass {W * H / best[i] / 1e3:9.1f} Mpaths/s  cull {gs[i][0].launch_info()['shadow_cull']}", flush=True)
            gs[i][0].close()
