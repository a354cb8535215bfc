"""Sum rocprofv3 --pmc CSV rows of the dominant radiance kernel (one directory per pass) and derive the
figures bench.py and DESIGN.md quote.  usage: summarize_pmc.py <pmc dir> <workload key> <render passes of the profiled run> [out.json [kernel_resources.json]]
(kernel_resources.json: scripts/kernel_resources.py - the compiler's own register / spill / scratch figures per kernel.
rocprofv3's VGPR_Count column counts in granules of 4 where gfx950's wave64 kernels are allotted granules of 8: it
reads 40 for this library's 80-register kernels; it is kept as `vgpr_count_rocprof` and the compiler's figure is `vgprs`.)

Derived (MI355X_MICROARCH.md, rocprofv3 PMC slots / HBM):
  active_lane_fraction    SQ_THREAD_CYCLES_VALU / (64 x SQ_INSTS_VALU)   (thread-cycles per issued wave instruction)
  occupancy_waves_per_cu  4 x SQ_WAVE_CYCLES / ((GRBM_GUI_ACTIVE / 8) x 256): SQ_WAVE_CYCLES counts quad-cycles of resident
                          waves, GRBM_GUI_ACTIVE is summed over the 8 XCDs - mean waves resident per CU while the kernel
                          runs (SQ_LEVEL_WAVES reads 0 on this image)
  FETCH_SIZE / WRITE_SIZE are in KB; FETCH_SIZE is doubled (gfx950 tallies the 128-B requests of 16-B-per-lane
                          reads at 64 B: the sample buffer is read as one float4 per lane), WRITE_SIZE taken as is
"""
import csv, glob, json, os, sys
from collections import defaultdict

root, key, passes = sys.argv[1], sys.argv[2], int(sys.argv[3])
tot = defaultdict(lambda: defaultdict(float))
disp = defaultdict(lambda: defaultdict(set))
for f in glob.glob(os.path.join(root, "*", "**", "*counter_collection.csv"), recursive=True):
    pass_name = os.path.relpath(f, root).split(os.sep)[0]
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"]
        if "radiance" not in k:
            continue
        tot[k][row["Counter_Name"]] += float(row["Counter_Value"])
        disp[k][pass_name].add(row["Dispatch_Id"])
        tot[k]["_vgpr"] = float(row["VGPR_Count"]); tot[k]["_sgpr"] = float(row["SGPR_Count"]); tot[k]["_lds"] = float(row["LDS_Block_Size"])
        tot[k]["_scratch"] = float(row["Scratch_Size"]); tot[k]["_wg"] = float(row["Workgroup_Size"])
if not tot:
    sys.exit("no radiance kernel rows under " + root)
kernel = max(tot, key=lambda k: tot[k].get("SQ_INSTS_VALU", 0) + tot[k].get("SQ_WAVE_CYCLES", 0))
c = tot[kernel]
n_disp = max(len(v) for v in disp[kernel].values())
g = lambda n: c.get(n)
# `passes` = render passes of the whole profiled run (scripts/profile_r02.sh: 2), however many dispatches they took
out = {"kernel": kernel, "dispatches_per_pmc_pass": n_disp, "passes": passes,
       "vgpr_count_rocprof": g("_vgpr"), "sgpr_count_rocprof": g("_sgpr"), "lds_block_bytes": g("_lds"),
       "scratch_bytes_per_lane": g("_scratch"), "workgroup_size": g("_wg")}
if len(sys.argv) > 5 and os.path.exists(sys.argv[5]):   # the compiler's figures for this kernel (scripts/kernel_resources.py)
    res = json.load(open(sys.argv[5]))
    short = kernel.split("(")[0].strip()
    short = short[5:] if short.startswith("void ") else short
    for name, v in res.items():
        if name == short:
            out.update({k: v[k] for k in ("vgprs", "sgprs", "sgpr_spill", "vgpr_spill", "waves_per_simd") if k in v})
            out["scratch_bytes_per_lane_compiler"] = v.get("scratch_bytes_per_lane")
for name in sorted(c):
    if not name.startswith("_"):
        out[name] = c[name]
if g("SQ_INSTS_VALU") and g("SQ_THREAD_CYCLES_VALU"):
    out["active_lane_fraction"] = g("SQ_THREAD_CYCLES_VALU") / (64 * g("SQ_INSTS_VALU"))
if g("SQ_WAVE_CYCLES") and g("GRBM_GUI_ACTIVE"):
    out["occupancy_waves_per_cu"] = 4 * g("SQ_WAVE_CYCLES") / ((g("GRBM_GUI_ACTIVE") / 8) * 256)
if g("SQ_WAVE_CYCLES"):
    for n in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_LDS"):
        if g(n) is not None:
            out[n + "_share_of_wave_cycles"] = g(n) / g("SQ_WAVE_CYCLES")
if g("SQ_LDS_BANK_CONFLICT") is not None and g("SQ_LDS_IDX_ACTIVE"):
    out["lds_bank_conflict_share"] = g("SQ_LDS_BANK_CONFLICT") / g("SQ_LDS_IDX_ACTIVE")
if g("SQ_INSTS_VALU"):
    fp = sum(g(n) or 0 for n in ("SQ_INSTS_VALU_FMA_F32", "SQ_INSTS_VALU_ADD_F32", "SQ_INSTS_VALU_MUL_F32", "SQ_INSTS_VALU_TRANS_F32"))
    out["fp32_share_of_valu"] = fp / g("SQ_INSTS_VALU")
    out["int32_share_of_valu"] = (g("SQ_INSTS_VALU_INT32") or 0) / g("SQ_INSTS_VALU")
    out["salu_per_valu"] = ((g("SQ_INSTS_SALU") or 0)) / g("SQ_INSTS_VALU")
if g("FETCH_SIZE") is not None:
    out["FETCH_SIZE_bytes_per_pass"] = 2 * g("FETCH_SIZE") * 1024 / out["passes"]
    out["FETCH_SIZE_raw_kb"] = g("FETCH_SIZE")
if g("WRITE_SIZE") is not None:
    out["WRITE_SIZE_bytes_per_pass"] = g("WRITE_SIZE") * 1024 / out["passes"]
print(json.dumps({key: out}, indent=1))
if len(sys.argv) > 4:
    path = sys.argv[4]
    try:
        allk = json.load(open(path))
    except (OSError, ValueError):
        allk = {}
    allk[key] = out
    json.dump(allk, open(path, "w"), indent=1, sort_keys=True)
