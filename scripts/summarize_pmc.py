"""Sum rocprofv3 --pmc CSV rows per kernel and counter (one directory per pass)."""
import csv, glob, os, sys
from collections import defaultdict

root = sys.argv[1]
tot = defaultdict(lambda: defaultdict(float))
calls = defaultdict(int)
for f in glob.glob(os.path.join(root, "*", "**", "*counter_collection.csv"), recursive=True):
    seen = set()
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"]
        if "radiance" not in k:
            continue
        tot[k][row["Counter_Name"]] += float(row["Counter_Value"])
        seen.add((k, row["Dispatch_Id"]))
    for k, _ in seen:
        calls[(k, f)] += 1
for k, c in tot.items():
    print(k)
    n = max(v for (kk, _), v in calls.items() if kk == k)
    print("  dispatches per pass:", n)
    for name in sorted(c):
        print(f"  {name:28s} {c[name]:.6g}")
    g = c.get
    if g("SQ_INSTS_VALU") and g("SQ_THREAD_CYCLES_VALU") and g("SQ_ACTIVE_INST_VALU"):
        # THREAD_CYCLES_VALU counts active lanes x cycles; ACTIVE_INST_VALU counts (quad-)cycles of VALU issue
        print("  active-lane fraction (THREAD_CYCLES_VALU / (64 * ACTIVE_INST_VALU*4)):",
              g("SQ_THREAD_CYCLES_VALU") / (64 * 4 * g("SQ_ACTIVE_INST_VALU")))
    if g("SQ_ACTIVE_INST_VALU") and g("SQ_BUSY_CYCLES"):
        print("  VALU busy (ACTIVE_INST_VALU*4 / BUSY_CYCLES / ...): see DESIGN.md")
