"""pyptgpu.gpu(): one HIP runtime per process (last paragraph)."""
import os


def test_a_second_hip_runtime_in_the_process_is_refused():
    """pyptgpu.gpu(): libptgpu.so loaded first binds ROCm's libamdhip64, torch imported afterwards
    maps its own copy - two runtimes; the next gpu() / Gpu() call refuses and names the import order.  In the right
    order (torch first) there is one copy and nothing is refused.  (Mapping only: no GPU is touched.)"""
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    code = ("import sys; sys.path.insert(0, %r)\n"
            "import pyptgpu as pt\n"
            "%s\n"
            "try:\n"
            "    pt.gpu(); print('loaded', len(pt.hip_runtimes_mapped()))\n"
            "except pt.GpuError as e:\n"
            "    print('refused:', e)\n") % (os.path.join(os.path.dirname(here), "path-tracer_amd"), "%s")
    wrong = subprocess.run([sys.executable, "-c", code % "pt.gpu(); import torch"], capture_output=True, text=True, timeout=300)
    assert "refused: two HIP runtimes" in wrong.stdout and "import torch" in wrong.stdout, wrong.stdout + wrong.stderr
    right = subprocess.run([sys.executable, "-c", code % "import torch"], capture_output=True, text=True, timeout=300)
    assert right.stdout.strip() == "loaded 1", right.stdout + right.stderr
