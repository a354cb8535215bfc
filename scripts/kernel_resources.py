"""What the compiler allotted to every kernel of libptgpu.so: registers, spills, scratch, occupancy - from hipcc's own
-Rpass-analysis=kernel-resource-usage remarks, one device-only compile per translation unit with the Makefile's flags.
usage: kernel_resources.py [out.json]   (default: stdout).  scripts/summarize_pmc.py merges the file into its summaries:
rocprofv3's VGPR_Count field is in other units (it reports 40 for this library's 80-register kernels)."""
import json, os, re, subprocess, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(REPO, "path-tracer_amd", "csrc")
COMMON = "--offload-arch=gfx950 -O3 -std=c++17 -fPIC -I../../include -I. -Wall -Wno-unused-function -fno-slp-vectorize -DPT_WF_WAVES_PER_SIMD=6".split()
out = {}
for tu in ("kernels_fast", "kernels_strict", "kernels_fast_bvh", "kernels_strict_bvh", "kernels_fast_hyb", "kernels_fast_ball", "kernels_strict_ball"):
    contract = "-ffp-contract=off" if "strict" in tu else "-ffp-contract=fast-honor-pragmas"
    cmd = ["/opt/rocm/bin/hipcc", *COMMON, contract, "--cuda-device-only", "-Rpass-analysis=kernel-resource-usage", "-c", tu + ".hip", "-o", "/dev/null"]
    text = subprocess.run(cmd, cwd=SRC, capture_output=True, text=True).stderr
    cur = None
    for line in text.splitlines():
        m = re.search(r"remark: Function Name: (\S+)", line)
        if m:
            cur = out.setdefault(m.group(1), {"translation_unit": tu})
            continue
        m = re.search(r"remark:\s+([A-Za-z ]+?)(?: \[[^\]]*\])?: (\d+)", line)
        if m and cur is not None:
            key = {"TotalSGPRs": "sgprs", "VGPRs": "vgprs", "AGPRs": "agprs", "ScratchSize": "scratch_bytes_per_lane", "Occupancy": "waves_per_simd",
                   "SGPRs Spill": "sgpr_spill", "VGPRs Spill": "vgpr_spill", "LDS Size": "static_lds_bytes", "Dynamic Stack": "dynamic_stack"}.get(m.group(1).strip())
            if key:
                cur[key] = int(m.group(2))
def pretty(mangled):   # _ZN<n>ns<n>kernelILb1EEEv... -> ns::kernel<true>  (enough for this library's kernels)
    m = re.match(r"_ZN(\d+)", mangled)
    if not m:
        return mangled
    i, parts = 3, []
    while i < len(mangled) and mangled[i].isdigit():
        j = i
        while mangled[j].isdigit():
            j += 1
        n = int(mangled[i:j])
        parts.append(mangled[j:j + n])
        i = j + n
    name = "::".join(parts)
    t = re.match(r"ILb([01])E", mangled[i:])
    return name + ("<true>" if t and t.group(1) == "1" else "<false>" if t else "")


result = {pretty(k): dict(v, mangled=k) for k, v in out.items()}
text = json.dumps(result, indent=1, sort_keys=True)
if len(sys.argv) > 1:
    open(sys.argv[1], "w").write(text + "\n")
else:
    print(text)
