"""Fast build (per cull mode) against the strict build (= the reference, bit for bit) on cornell.txt:
mean bias per channel, RMSE, block-averaged relative error, NaN pixels.  Run on the GPU box."""
import sys, os, json
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "path-tracer_amd"))
import pyptgpu as pt

def render(sc, w, h, spp, math, cull, seeds):
    g = pt.Gpu(0, 0, math=math, shadow_cull=cull)
    g.set_scene(sc); g.resize(w, h, seeds); g.render(spp); g.finish()
    out = g.colors(); g.close(); return out

def stats(a, b, w, h, block=8):
    ok = np.isfinite(a).all(1) & np.isfinite(b).all(1)
    d = a[ok].astype(np.float64) - b[ok]
    m = b[ok].astype(np.float64).mean(0)
    res = {"mean_rel_bias_rgb": (d.mean(0) / m).tolist(), "rmse": float(np.sqrt((d ** 2).mean())),
           "rel_rmse": float(np.sqrt((d ** 2).mean()) / b[ok].mean()),
           "share_gt_1e-4": float((np.abs(d).max(1) > 1e-4).mean()),
           "nan_a": int((~np.isfinite(a).all(1)).sum()), "nan_b": int((~np.isfinite(b).all(1)).sum())}
    A = np.where(ok[:, None], a, 0).reshape(h, w, 3).astype(np.float64); B = np.where(ok[:, None], b, 0).reshape(h, w, 3).astype(np.float64)
    hb, wb = h // block, w // block
    Ab = A[:hb * block, :wb * block].reshape(hb, block, wb, block, 3).mean((1, 3, 4)); Bb = B[:hb * block, :wb * block].reshape(hb, block, wb, block, 3).mean((1, 3, 4))
    rel = np.abs(Ab - Bb) / np.maximum(Bb, 1e-3)
    res["block_rel_err_max"] = float(rel.max()); res["block_rel_err_p99"] = float(np.quantile(rel, .99)); res["block_rel_err_mean"] = float(rel.mean())
    return res

out = {}
quick = len(sys.argv) > 2 and sys.argv[2] == "quick"
for (w, h, spp) in (((1024, 768, 4),) if quick else ((256, 192, 64), (1024, 768, 4))):
    sc = pt.LoadedScene("cornell", w, h); seeds = pt.seed_block(w, h)
    strict = render(sc, w, h, spp, pt.MATH_STRICT, pt.CULL_SAFE, seeds)
    for name, cull in ((("safe", pt.CULL_SAFE),) if quick else (("safe", pt.CULL_SAFE), ("tight", pt.CULL_TIGHT), ("enclosed", pt.CULL_ENCLOSED), ("ball", pt.CULL_BALL))):
        fast = render(sc, w, h, spp, pt.MATH_FAST, cull, seeds)
        out[f"{w}x{h}x{spp} fast+{name} vs strict"] = stats(fast, strict, w, h)
        print(f"{w}x{h}x{spp} fast+{name}", json.dumps(out[f"{w}x{h}x{spp} fast+{name} vs strict"]), flush=True)
    if quick:
        continue
    st2 = render(sc, w, h, spp, pt.MATH_STRICT, pt.CULL_ENCLOSED, seeds)
    out[f"{w}x{h}x{spp} strict+enclosed vs strict"] = stats(st2, strict, w, h)
    print(f"{w}x{h}x{spp} strict+enclosed", json.dumps(out[f"{w}x{h}x{spp} strict+enclosed vs strict"]), flush=True)
json.dump(out, open(sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/fast_vs_strict.json", "w"), indent=1)
