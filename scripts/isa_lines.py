"""Static instruction mix of one kernel from line-annotated ISA (hipcc -S -gline-tables-only): per basic block (with the
loop depth the compiler prints) and per source line, VALU / SALU / LDS / VMEM / SMEM / spill counts.
usage: isa_lines.py file.s <kernel name substring> [--blocks] [--lines]"""
import sys, re, collections
path, pat = sys.argv[1], sys.argv[2]
want_blocks = "--blocks" in sys.argv
want_lines = "--lines" in sys.argv or not want_blocks
files = {}
inside = False
loc = ("?", 0)
block = "entry"
depth = 0
per_line = collections.defaultdict(collections.Counter)
per_block = collections.OrderedDict()
def cls(op):
    if op.startswith(("v_writelane", "v_readlane")): return "spill?"
    if op.startswith("v_"): return "valu"
    if op.startswith(("s_load", "s_buffer_load")): return "smem"
    if op.startswith(("s_waitcnt", "s_nop")): return "wait"
    if op.startswith(("s_cbranch", "s_branch")): return "branch"
    if op.startswith("s_"): return "salu"
    if op.startswith("ds_"): return "lds"
    if op.startswith(("global_", "flat_", "buffer_")): return "vmem"
    if op.startswith("scratch_"): return "scratch"
    return "other"
for raw in open(path):
    line = raw.rstrip("\n")
    m = re.match(r"\s*\.file\s+(\d+)\s+\"([^\"]*)\"\s+\"([^\"]*)\"", line)
    if m:
        files[int(m.group(1))] = m.group(3)
        continue
    m = re.match(r"^(_Z\w+):", line)
    if m:
        inside = pat in m.group(1)
        block, depth = "entry", 0
        continue
    if not inside:
        continue
    if re.match(r"^\.Lfunc_end", line):
        inside = False
        continue
    m = re.match(r"\s*\.loc\s+(\d+)\s+(\d+)", line)
    if m:
        loc = (files.get(int(m.group(1)), "?"), int(m.group(2)))
        continue
    m = re.match(r"^(\.LBB\d+_\d+):(.*)", line)
    if m:
        block = m.group(1)
        d = re.search(r"Depth=(\d+)", m.group(2))
        depth = int(d.group(1)) if d else 0
        per_block[block] = {"depth": depth, "n": collections.Counter(), "lines": collections.Counter()}
        continue
    d = re.search(r";\s+=>\s+This (?:Inner )?Loop Header: Depth=(\d+)", line)
    if d and block in per_block:
        per_block[block]["depth"] = int(d.group(1))
        continue
    m = re.match(r"^\s+([a-z_0-9]+)\b", line)
    if not m or line.strip().startswith((";", ".")):
        continue
    c = cls(m.group(1))
    per_line[loc][c] += 1
    if block not in per_block:
        per_block[block] = {"depth": depth, "n": collections.Counter(), "lines": collections.Counter()}
    per_block[block]["n"][c] += 1
    per_block[block]["lines"][loc] += 1
keys = ["valu", "salu", "branch", "wait", "lds", "smem", "vmem", "scratch", "spill?"]
if want_blocks:
    print("block depth " + " ".join(f"{k:>7s}" for k in keys) + "  main source lines")
    for b, v in per_block.items():
        top = ", ".join(f"{f.split('_')[-1].replace('.inc','')}:{l}x{n}" for (f, l), n in v["lines"].most_common(4))
        print(f"{b:12s} {v['depth']:2d} " + " ".join(f"{v['n'][k]:7d}" for k in keys) + "  " + top)
if want_lines:
    tot = collections.Counter()
    rows = []
    for (f, l), c in per_line.items():
        rows.append((sum(c.values()), f, l, c))
        tot.update(c)
    print("total", dict(tot))
    for n, f, l, c in sorted(rows, reverse=True)[:120]:
        print(f"{n:6d} {f}:{l:<5d} " + " ".join(f"{k}={c[k]}" for k in keys if c[k]))
