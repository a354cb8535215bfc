"""Copies what scripts/r03_configs.sh left under gpurun_out/ into profiles/r03/ (tracked): bench lines, kernel
stats of the rocprofv3 traces, and the PMC summaries merged key by key into profiles/r03/pmc_summary.json."""
import glob, json, os, shutil
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
dst = os.path.join(R, "profiles", "r03")
os.makedirs(os.path.join(dst, "bench"), exist_ok=True)
try:
    summ = json.load(open(os.path.join(dst, "pmc_summary.json")))
except (OSError, ValueError):
    summ = {}
for d in sorted(glob.glob(os.path.join(R, "gpurun_out", "prof_r03", "*", ""))):
    key = os.path.basename(d.rstrip("/"))
    if os.path.exists(os.path.join(d, "kernel_stats.csv")):
        shutil.copy(os.path.join(d, "kernel_stats.csv"), os.path.join(dst, "kernel_stats_%s.csv" % key))
    try:
        summ.update(json.loads(open(os.path.join(d, "pmc_summary.txt")).read()))
    except (OSError, ValueError) as e:
        print("no PMC summary for", key, e)
json.dump(summ, open(os.path.join(dst, "pmc_summary.json"), "w"), indent=1, sort_keys=True)
for f in glob.glob(os.path.join(R, "gpurun_out", "bench_r03", "*.json")):
    if os.path.getsize(f):
        shutil.copy(f, os.path.join(dst, "bench", os.path.basename(f)))
print(sorted(summ))
