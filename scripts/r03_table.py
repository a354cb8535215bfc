"""Prints DESIGN.md's measurement tables (sections 5.3 and 5.5) from the tracked files under profiles/r03/:
every cell is a field of a bench line (profiles/r03/bench/*.json) or of profiles/r03/pmc_summary.json."""
import json, os
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
B = os.path.join(R, "profiles", "r03", "bench")
def line(name):
    return json.loads(open(os.path.join(B, name + ".json")).read().strip().splitlines()[-1])
def num(x, digits=3):
    return "-" if x is None else f"{x:.{digits}f}"
print("| config | Mpaths/s | instance / kernel | executed flop/path | frac | issued (PMC) | active lanes | waves/CU | reference-equivalent flop/path | CPU 16 thr / 1 thr (Mpaths/s) | bias / rel. RMSE vs strict | NaN px fast/strict |")
print("|---|---|---|---|---|---|---|---|---|---|---|---|")
for name, label in (("cornell_pt", "2: cornell pt"), ("materials_pt", "3: materials pt"), ("materials_dl", "3: materials dl"),
                    ("caustic_pt", "3: caustic pt"), ("caustic_dl", "3: caustic dl"), ("obj-model_pt", "4: obj-model pt (BVH)"),
                    ("obj-model-bunny_pt", "4: obj-model-bunny pt (BVH, nodes in LDS)"), ("cornell-large_4k_pt", "5: cornell-large pt, 3840x2160")):
    d = line(name); r = d["roofline"]; v = r.get("valu_fp32_issued") or {}; e = d.get("image_error") or {}; c = d.get("cpu_baseline") or {}
    kern = r["kernel"].replace("ptk_", "").replace("::radiance_", " / ").replace("_kernel", "")
    print(f"| {label} | {d['value']:.0f} | {kern} | {r['algorithmic_flops_per_path']:.0f} | {r['frac']:.3f} | {num(v.get('frac'))} | "
          f"{num(v.get('active_lane_fraction'), 2)} | {num(v.get('occupancy_waves_per_cu'), 1)} | {r.get('reference_equivalent_flops_per_path', 0):.0f} | "
          f"{num(c.get('value'), 3)} / {num(c.get('single_thread_value'), 4)} | {e.get('mean_rel_bias', 0):+.1e} / {100 * e.get('rel_rmse', 0):.2f} % | "
          f"{e.get('nan_pixels_fast')} / {e.get('nan_pixels_strict')} |")
print()
print("| build | sets | Mpaths/s | executed kflop/path | frac | mean bias vs strict |")
print("|---|---|---|---|---|---|")
for name, label in (("cornell_pt", "fast | ENCLOSED (default)"), ("cornell_cull_ball", "fast | BALL (opt-in)"), ("cornell_cull_tight", "fast | TIGHT"),
                    ("cornell_cull_safe", "fast | SAFE = the reference's algorithm, fast arithmetic"), ("cornell_strict_safe", "strict | SAFE = the reference bit for bit, its own scan order"),
                    ("cornell_strict", "strict | none: every ray through the hierarchy (strict default) = the reference bit for bit")):
    d = line(name); r = d["roofline"]; e = d.get("image_error") or {}
    b = e.get("mean_rel_bias")
    print(f"| {label} | {d['value']:.0f} | {r['algorithmic_flops_per_path'] / 1e3:.1f} | {r['frac']:.3f} | {'0' if b is None else f'{b:+.1e}'} |")
