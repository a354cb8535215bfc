#!/usr/bin/env python3
"""bench.py - Mpaths/s of the radiance path; default workload scenes/cornell.txt 1024x768, path tracing
(BASELINE.json config 2).  Other configs: --scene materials|caustic|obj-model|obj-model-bunny|cornell-large,
--integrator pt|dl, --width/--height.

    python bench.py --gpus N --steps K --warmup W

A step = `--passes` progressive passes (one path per pixel per pass) over the whole frame, issued through the
C-ABI as ONE render call (for the default workload the library makes it one launch of the work-queue kernel).
The defaults time 16 x 64 = 1 024 spp, the sample count BASELINE.json's config names.  Inputs (scene records,
seeds, sample buffer) are resident in HBM before the timed region; the region is K steps bracketed by a barrier
+ device sync, MAX over ranks.  With N > 1 each rank renders from its own seed block and the sample buffers are
summed with one RCCL all-reduce inside the timed region (weak scaling: per-GPU work is fixed).  Started WITHOUT
a launcher and with --gpus N > 1, it starts its own N ranks (torch.distributed.run) before touching the GPU.

The JSON line also carries
  roofline     bound FP32 VALU.  `achieved` = flops the kernel EXECUTES per path x paths per
               step / the step's HIP-event time.  The executed work is counted by the kernel itself: the
               counting build of the library (-DPT_COUNT_WORK=1, same source, one atomic per wave and event)
               renders a reduced frame of the same scene with the same options outside the timed region and
               reports primitive tests, hierarchy node tests, light samples ... per ACTIVE lane; each is
               weighted with its flop count (FLOPS below, FMA = 2).  `valu_fp32_issued` is the second,
               independent figure: FP32 VALU instructions the hardware issued (committed rocprofv3 PMC run of
               the same workload, profiles/r02) x 64 lanes x the measured active-lane share / this run's step
               time.  The reference's brute-force count is reported beside them as reference_equivalent_* and
               is NOT what `frac` is built from.
  image_error  the benchmarked (fast) build against the strict build - which equals the reference bit for bit
               (tests) - same seeds, same spp, full frame: RMSE, mean bias, NaN pixels.
  exact_algorithm  secondary rates of the same workload (not `value`): fast math + SAFE sets, fast math +
               ENCLOSED_SELF (= TIGHT's image bit for bit in the strict build) and the strict build's default.
  cpu_baseline the CPU oracle (a bit-exact port of the reference's main-cpu.c path) timed on this host's cores
               on a bounded sample of the same scene.
"""
import argparse
import json
import os
import subprocess
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(REPO, "path-tracer_amd"))

PEAK_FP32_VALU_TFLOPS = 157.3  # MI355X_MICROARCH.md: 256 CU x 128 lanes x 2 x 2.4 GHz

# flops per executed event (FMA = 2 flops; compares and selects not counted), DESIGN.md 5
FLOPS = {"sphere_tests": 20,            # geomfunc.h:10-39: 3 sub, 2 dots (10), b*b - . + rr (3), sqrt, 2 add/sub
         "triangle_tests_mt": 52,       # geomfunc.h:45-78: 2 cross, 4 dot, rcp, 3 mul, 1 add
         "triangle_tests_planes": 32,   # triangle_planes_t: 4 dots of 3 (20), rcp, sub, mul, P (6), 2 add, 1 add
         "bvh_nodes": 22,               # slab test: 6 sub, 6 mul, 10 min/max
         "light_samples": 70,           # geomfunc.h:227-300 without the shadow scan
         "plane_side_tests": 5,         # dot of 3 against the light's plane
         "segment_near_tests": 21,      # 3 sub, dot, med3, 3 fma, dot, mul
         "ball_point_tests": 8,         # 3 sub, dot of 3
         "convex_plane_tests": 13,      # 2 dots of 3, sub, rcp, mul (segment_crosses_convex)
         "extension_rays": 120}         # shading of a closest hit, beside its primitive tests
# a light point sampled once more for a binned ray (wf_trace_near): sqrt, 2 scales, 2 adds, sub, dot, rsq, scale
FLOPS_RESAMPLED_POINT = 33


def spawn_ranks(n):
    """Started without a launcher: run the same command under torch.distributed.run, one rank per GPU.
    Nothing in this process has touched the GPU yet."""
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.run(cmd).returncode


def cpu_threads(asked):
    """Threads of the CPU leg: every core this process may run on.  The GPU pool allots a share of the machine to
    a one-GPU box and states it in OMP_NUM_THREADS (16; os.cpu_count() there shows the whole machine): that share
    is taken when set, otherwise the affinity mask.  --cpu-threads overrides; os.cpu_count() is printed either way."""
    if asked > 0:
        return asked
    env = int(os.environ.get("OMP_NUM_THREADS", "0") or 0)
    if env > 0:
        return env
    try:
        return len(os.sched_getaffinity(0))
    except AttributeError:
        return os.cpu_count() or 1


def cpu_baseline(scene_name, threads_asked=0):
    """Rank 0, N = 1 only.  oracle/ is used here as the timed CPU baseline and the reference's flop counter."""
    sys.path.insert(0, os.path.join(REPO, "oracle"))
    import oracle_binding as ob
    import pyptgpu as pt
    threads = cpu_threads(threads_asked)
    # flops per path of the reference algorithm (full scans), counted on a 96x72 frame (serial, counting build)
    cw, ch = 96, 72
    sc = pt.LoadedScene(scene_name, cw, ch)
    ctr = ob.Counters()
    ob.Render(sc, pt.seed_block(cw, ch)).pass_oracle_counted(ctr, 1)
    # single thread = the reference's serial loop (main-cpu.c:61-62), at least ~1 s of it
    t_one, passes_one = 0.0, 0
    r = ob.Render(sc, pt.seed_block(cw, ch))
    while t_one < 1.0 and passes_one < 64:
        t0 = time.perf_counter()
        r = r.pass_oracle(1, threads=1)
        t_one += time.perf_counter() - t0
        passes_one += 1
    one = cw * ch * passes_one / t_one / 1e6
    # all threads, sized for ~10-20 s of wall time
    paths_target = max(one * 1e6 * threads * 12.0, cw * ch)
    scale = max(1, int((paths_target / (cw * ch)) ** 0.5))
    bw, bh = min(1024, cw * scale) // 8 * 8, min(768, ch * scale) // 8 * 8
    spp = max(1, min(64, int(paths_target / (bw * bh))))
    scb = pt.LoadedScene(scene_name, bw, bh)
    t0 = time.perf_counter()
    ob.Render(scb, pt.seed_block(bw, bh)).pass_oracle(spp, threads=threads)
    t_all = time.perf_counter() - t0
    return {
        "value": bw * bh * spp / t_all / 1e6, "unit": "Mpaths/s", "cores": threads, "kind": "port",
        "sample": f"{scene_name} {bw}x{bh} x {spp} spp, {threads} OpenMP threads over rows, {t_all:.1f} s",
        "host_cpu_count": os.cpu_count(), "affinity_cpus": len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else None,
        "omp_num_threads_env": os.environ.get("OMP_NUM_THREADS"),
        "cores_note": "cores = OMP_NUM_THREADS when the pool sets it (its CPU share of the machine per GPU), else the affinity mask; "
                      "--cpu-threads N overrides.  The oracle scales with rows: value x host_cpu_count / cores bounds the whole machine",
        "single_thread_value": one,
        "single_thread_sample": f"{scene_name} {cw}x{ch} x {passes_one} spp, serial loop as main-cpu.c, {t_one:.1f} s",
    }, ctr.flops() / ctr.paths, ctr.as_dict()


def executed_work(pt, args, make_gpu, w, h):
    """Counting build, reduced frame, same scene and options: events per path of what the kernel executes."""
    scale = max(w / 256.0, h / 192.0, 1.0)
    cw, ch = max(8, int(w / scale) // 8 * 8), max(8, int(h / scale) // 8 * 8)
    passes = 2
    sc = pt.LoadedScene(args.scene, cw, ch)
    g = make_gpu(lib=pt.variant_path("count"))
    g.set_scene(sc)
    g.resize(cw, ch, pt.seed_block(cw, ch))
    g.work_counters(reset=True)
    g.render(passes)
    g.finish()
    work = g.work_counters()
    li = g.launch_info()
    g.close()
    paths = work["paths"]
    per_path = {k: v / paths for k, v in work.items()}
    flops = sum(FLOPS[k] * per_path[k] for k in FLOPS)
    # light points sampled beyond the light samples themselves = binned rays rebuilt when traced
    flops += FLOPS_RESAMPLED_POINT * max(0.0, per_path["light_points"] - per_path["light_samples"])
    return flops, per_path, f"{args.scene}.txt {cw}x{ch} x {passes} spp, counting build", li


def exact_algorithm_rates(pt, make_gpu, sc, seeds, w, h, passes):
    """Secondary figures, outside the timed region: the same workload through the builds that run the reference's
    algorithm with no candidate-set rule of this library's own - fast math + SAFE sets (statistical parity by the
    arithmetic alone) and the strict build with its default (bit-exact with the CPU path).  HIP-event times, best of 3."""
    rates = {}
    for name, math, cull in (("fast_safe", pt.MATH_FAST, pt.CULL_SAFE), ("fast_enclosed_self", pt.MATH_FAST, pt.CULL_ENCLOSED_SELF),
                             ("strict_default", pt.MATH_STRICT, pt.CULL_AUTO)):
        g = make_gpu(math=math, cull=cull)
        g.set_scene(sc)
        g.resize(w, h, seeds)
        n = passes if math == pt.MATH_FAST else max(1, passes // 4)
        g.render_timed(n)
        ms = min(g.render_timed(n) for _ in range(3))
        li = g.launch_info()
        g.close()
        rates[name] = {"Mpaths_per_s": w * h * n / ms / 1e3, "passes": n, "ms": ms, "shadow_cull": li["shadow_cull"],
                       "kernel": "wavefront" if li["kernel"] == pt.KERNEL_WAVEFRONT else "mega"}
    return rates


def image_error(pt, make_gpu, sc, seeds, w, h, spp):
    """fast build (as benchmarked) against the strict build (= the reference, bit for bit) at equal seeds/spp."""
    import numpy as np
    imgs = {}
    for name, math in (("fast", pt.MATH_FAST), ("strict", pt.MATH_STRICT)):
        g = make_gpu(math=math, cull=(None if name == "fast" else pt.CULL_AUTO))
        g.set_scene(sc)
        g.resize(w, h, seeds)
        g.render(spp)
        g.finish()
        imgs[name] = g.colors()
        g.close()
    a, b = imgs["fast"], imgs["strict"]
    ok = np.isfinite(a).all(1) & np.isfinite(b).all(1)
    d = a[ok].astype(np.float64) - b[ok]
    mean = float(b[ok].astype(np.float64).mean())
    return {"vs": "strict build (bit-exact with the reference's CPU path: tests/test_gpu_parity.py), same seeds",
            "frame": f"{w}x{h}", "spp": spp,
            "rmse": float(np.sqrt((d ** 2).mean())), "rel_rmse": float(np.sqrt((d ** 2).mean()) / mean),
            "mean_bias": float(d.mean()), "mean_rel_bias": float(d.mean() / mean),
            "mean_rel_bias_rgb": (d.mean(0) / b[ok].astype(np.float64).mean(0)).tolist(),
            "max_abs": float(np.abs(d).max()), "mean_radiance_strict": mean,
            "nan_pixels_fast": int((~np.isfinite(a).all(1)).sum()), "nan_pixels_strict": int((~np.isfinite(b).all(1)).sum())}


def committed_profile(workload_key):
    """Per-pass PMC figures of the newest committed rocprofv3 run of this workload (profiles/rNN/pmc_summary.json)."""
    for rnd in ("r03", "r02"):
        try:
            prof = json.load(open(os.path.join(REPO, "profiles", rnd, "pmc_summary.json"))).get(workload_key)
        except (OSError, ValueError):
            prof = None
        if prof:
            prof["round"] = rnd
            return prof
    return None


def keep_stdout_for_the_line():
    """The contract is ONE JSON line on stdout.  RCCL prints a version banner to the process's stdout when its
    first communicator comes up (and other libraries may print as they load): from here on file descriptor 1 is
    stderr, and the line goes to the descriptor returned, the stdout this process was started with."""
    sys.stdout.flush()
    real = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)
    return real


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=16)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--passes", type=int, default=64,
                    help="progressive passes per step (the reference's frame after warm-up is 0.5 s of passes, "
                         "main-gpu.c:774-794: hundreds at this rate); 16 steps x 64 = the 1 024 spp of BASELINE config 2")
    ap.add_argument("--scene", default="cornell")
    ap.add_argument("--width", type=int, default=1024)
    ap.add_argument("--height", type=int, default=768)
    ap.add_argument("--kernel", default="auto", choices=["auto", "mega", "wavefront"])
    ap.add_argument("--math", default="fast", choices=["fast", "strict"])
    ap.add_argument("--integrator", default="pt", choices=["pt", "dl"])
    ap.add_argument("--cull", default="auto", choices=["off", "safe", "tight", "enclosed", "self", "ball", "auto"],
                    help="shadow-ray candidate sets (auto = enclosed with fast math, safe with strict; ball is opt-in)")
    ap.add_argument("--streams", type=int, default=4, help="image regions on concurrent HIP streams (1..4; without the work queue)")
    ap.add_argument("--work-queue", default="auto", choices=["auto", "on", "off"],
                    help="wavefront kernel: one launch per step whose waves take (pass, tile) units (PTGPU_OPT_WORK_QUEUE)")
    ap.add_argument("--passes-per-launch", type=int, default=0, help="0 = automatic")
    ap.add_argument("--backend", default="nccl", help="process-group backend for N > 1 (nccl = RCCL; gloo to rehearse on one GPU)")
    ap.add_argument("--no-cpu", action="store_true", help="skip the CPU baseline leg")
    ap.add_argument("--cpu-threads", type=int, default=0, help="threads of the CPU baseline leg (0 = OMP_NUM_THREADS, else every core of the affinity mask)")
    ap.add_argument("--no-exact", action="store_true", help="skip the secondary timings of the exact-algorithm builds")
    ap.add_argument("--no-count", action="store_true", help="skip the counting-build run (roofline is then null)")
    ap.add_argument("--no-image-error", action="store_true", help="skip the fast-vs-strict comparison")
    ap.add_argument("--key-only", action="store_true", help="render one pass and print the workload key (scripts/profile_r02.sh)")
    args = ap.parse_args()

    if args.gpus > 1 and "RANK" not in os.environ:
        raise SystemExit(spawn_ranks(args.gpus))   # before anything touches the GPU

    import torch
    import multigpu
    import pyptgpu as pt

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus != world and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the radiance path has no CPU fallback")
    local_rank %= torch.cuda.device_count()  # rehearsals may put several ranks on one GPU (gloo only)
    torch.cuda.set_device(local_rank)
    w, h = args.width, args.height
    sc = pt.LoadedScene(args.scene, w, h)
    seeds = pt.seed_block(w, h, rank)
    culls = {"off": pt.CULL_OFF, "safe": pt.CULL_SAFE, "tight": pt.CULL_TIGHT, "enclosed": pt.CULL_ENCLOSED,
             "ball": pt.CULL_BALL, "self": pt.CULL_ENCLOSED_SELF, "auto": pt.CULL_AUTO}

    def make_gpu(math=None, cull=None, lib=None):
        return pt.Gpu(local_rank, pt.INTEGRATOR_PT if args.integrator == "pt" else pt.INTEGRATOR_DL,
                      kernel={"auto": pt.KERNEL_AUTO, "mega": pt.KERNEL_MEGA, "wavefront": pt.KERNEL_WAVEFRONT}[args.kernel],
                      math=(pt.MATH_FAST if args.math == "fast" else pt.MATH_STRICT) if math is None else math,
                      streams=args.streams, work_queue={"auto": 2, "on": 1, "off": 0}[args.work_queue],
                      passes_per_launch=args.passes_per_launch,
                      shadow_cull=culls[args.cull] if cull is None else cull, lib=lib)

    # The library's region streams are created BEFORE the process group: HIP spreads streams over
    # GPU_MAX_HW_QUEUES (4) hardware queues in creation order.
    out = keep_stdout_for_the_line()
    g = make_gpu()
    g.set_scene(sc)
    g.resize(w, h, seeds)
    dist = None
    if world > 1 or "RANK" in os.environ:   # under torchrun even a group of one goes through RCCL (rehearsal on a 1-GPU box)
        import torch.distributed as dist
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(args.backend)
    colors_t = torch.as_tensor(multigpu.DeviceArray(g.colors_device_ptr(), pt.COLOR_STRIDE_FLOATS * w * h),
                               device=torch.device("cuda", local_rank))

    def barrier():
        if dist is not None:
            dist.barrier()
        g.finish()
        torch.cuda.synchronize()

    if args.key_only:
        g.render(1)
        g.finish()
        li = g.launch_info()
        names = {v: k for k, v in culls.items() if k != "auto"}
        print(f"{args.scene}-{w}x{h}-{args.integrator}-{args.math}-{names[li['shadow_cull']]}-"
              f"{'wavefront' if li['kernel'] == pt.KERNEL_WAVEFRONT else 'mega'}", file=out, flush=True)
        g.close()
        return
    for _ in range(args.warmup):
        g.render_timed(args.passes)
    if dist is not None:
        # RCCL sets up channels and loads its kernels on the first collective of a given size: that
        # belongs to start-up, not to the timed region
        g.finish()
        warm = torch.zeros_like(colors_t)
        for _ in range(max(1, min(args.warmup, 2))):
            dist.all_reduce(warm, op=dist.ReduceOp.SUM)
        del warm
    barrier()
    launch_ms = []
    t0 = time.perf_counter()
    for _ in range(args.steps):
        launch_ms.append(g.render_timed(args.passes))  # HIP events on the library's stream
    g.finish()
    t1 = time.perf_counter()
    scale = multigpu.allreduce_mean_(colors_t, world)  # the one collective of the path
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    g.scale_and_repack(scale)
    g.finish()
    t3 = time.perf_counter()
    barrier()
    elapsed = time.perf_counter() - t0
    if os.environ.get("PT_BENCH_TRACE"):
        print(f"[rank {rank}] render {1e3 * (t1 - t0):.2f} ms, all-reduce {1e3 * (t2 - t1):.2f}, repack {1e3 * (t3 - t2):.2f}, "
              f"barrier {1e3 * (t0 + elapsed - t3):.2f}", file=sys.stderr, flush=True)
    ranks_seen = 1
    if dist is not None:
        t = torch.tensor([elapsed], device="cuda", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        ranks_seen = dist.get_world_size()   # from the communicator, not from the command line

    paths_per_step = w * h * args.passes
    total_paths = paths_per_step * args.steps * world
    value = total_paths / elapsed / 1e6

    if rank == 0:
        pixels = g.pixels()
        li = g.launch_info()
        cull_names = {v: k for k, v in culls.items() if k != "auto"}
        kernel_name = "radiance_%s_kernel" % ("wavefront" if li["kernel"] == pt.KERNEL_WAVEFRONT else "mega")
        instance = ("fast" if li["math"] == pt.MATH_FAST else "strict") + {0: "", 1: "_bvh", 2: "_hyb", 3: "_ball"}[li["instance"]]
        avg_step_ms = sum(launch_ms) / len(launch_ms)
        launches_per_step = 1 if li["work_queue"] else (args.passes + li["passes_per_launch"] - 1) // max(li["passes_per_launch"], 1)
        workload_key = (f"{args.scene}-{w}x{h}-{args.integrator}-{args.math}-{cull_names[li['shadow_cull']]}-"
                        f"{'wavefront' if li['kernel'] == pt.KERNEL_WAVEFRONT else 'mega'}")

        baseline = ref_flops_per_path = ref_counters = None
        if world == 1 and not args.no_cpu:
            baseline, ref_flops_per_path, ref_counters = cpu_baseline(args.scene, args.cpu_threads)
        roofline = None
        if not args.no_count:   # (per launch on one GPU: rank 0's kernel stands for every rank's at N > 1)
            flops_per_path, per_path, count_sample, count_li = executed_work(pt, args, make_gpu, w, h)
            achieved = flops_per_path * paths_per_step / (avg_step_ms * 1e-3) / 1e12
            prof = committed_profile(workload_key)
            issued = None
            traffic = None
            if prof:
                # FP32 VALU work the hardware issued per pass (wave instructions x 64 lanes x active share)
                fl = (2 * prof["SQ_INSTS_VALU_FMA_F32"] + prof["SQ_INSTS_VALU_ADD_F32"] + prof["SQ_INSTS_VALU_MUL_F32"]
                      + prof["SQ_INSTS_VALU_TRANS_F32"]) * 64 * prof["active_lane_fraction"] / prof["passes"]
                tf = fl * args.passes / (avg_step_ms * 1e-3) / 1e12
                issued = {"tflops": tf, "frac": tf / PEAK_FP32_VALU_TFLOPS, "flops_per_path": fl / (w * h),
                          "active_lane_fraction": prof["active_lane_fraction"],
                          "occupancy_waves_per_cu": prof.get("occupancy_waves_per_cu"),
                          "source": "profiles/%s/pmc_summary.json[%s] (rocprofv3 --pmc, separate passes) x this run's step time" % (prof["round"], workload_key)}
                if prof.get("FETCH_SIZE_bytes_per_pass") is not None and prof.get("WRITE_SIZE_bytes_per_pass") is not None:
                    traffic = (prof["FETCH_SIZE_bytes_per_pass"] + prof["WRITE_SIZE_bytes_per_pass"]) * args.passes
            roofline = {"bound": "valu_fp32", "achieved": achieved, "peak": PEAK_FP32_VALU_TFLOPS, "unit": "TFLOP/s",
                        "frac": achieved / PEAK_FP32_VALU_TFLOPS, "traffic": traffic,
                        "traffic_unit": "HBM bytes per step (PMC FETCH_SIZE + WRITE_SIZE of the committed run), algorithmic %d" % (52 * w * h * args.passes),
                        "kernel": f"ptk_{instance}::{kernel_name}",
                        "avg_step_event_ms": avg_step_ms, "launches_per_step": launches_per_step,
                        "avg_launch_ms": avg_step_ms / launches_per_step,
                        "concurrent_region_streams": li["regions"],
                        "work_queue": bool(li["work_queue"]), "unit_passes": li["unit_passes"],
                        "passes_per_launch": li["passes_per_launch"], "lds_bytes_staged": li["lds_bytes"],
                        "algorithmic_flops_per_path": flops_per_path,
                        "algorithm": "work the kernel executes, counted by the kernel (counting build) per active lane: "
                                     "events per path x flops per event",
                        "executed_events_per_path": per_path, "flops_per_event": FLOPS, "count_sample": count_sample,
                        "count_sample_launch_matches": {k: count_li[k] == li[k] for k in ("kernel", "instance", "shadow_cull", "ray_bins")},
                        "valu_fp32_issued": issued,
                        "reference_equivalent_flops_per_path": ref_flops_per_path,
                        "reference_equivalent_tflops": (ref_flops_per_path * paths_per_step / (avg_step_ms * 1e-3) / 1e12)
                                                       if ref_flops_per_path else None,
                        "reference_equivalent_note": "the reference's brute-force scans (SURVEY 8-d formula on the CPU oracle's "
                                                     "counters); NOT what frac is built from",
                        "paths_per_step": paths_per_step, "hbm_algorithmic_bytes_per_pass": 52 * w * h}
        err = None
        if world == 1 and args.math == "fast" and not args.no_image_error:
            err = image_error(pt, make_gpu, sc, seeds, w, h, min(16, args.passes))
        exact = None
        if world == 1 and args.math == "fast" and not args.no_exact:
            exact = exact_algorithm_rates(pt, make_gpu, sc, seeds, w, h, args.passes)
        line = {
            "metric": "Mpaths/s", "value": value, "unit": "Mpaths/s", "n_gpus": world, "ranks_seen": ranks_seen,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
            "data": "scene file data/scenes/%s.txt (the reference's own), rand()-compatible seeds" % args.scene,
            "config": {"workload": f"{args.scene}.txt {w}x{h}, {args.steps * args.passes} spp timed ({args.passes} passes/step), "
                                   f"{args.integrator} integrator, {kernel_name} ({instance}), shadow cull {cull_names[li['shadow_cull']]}",
                       "spp_timed": args.steps * args.passes, "spp_per_s": args.steps * args.passes * world / elapsed,
                       "parallelism": f"passes split over {world} GPU(s) by seed block + 1 all-reduce of the float4 sample buffer "
                                      f"per {args.steps * args.passes} passes per rank"},
            "workload_key": workload_key, "roofline": roofline, "cpu_baseline": baseline, "image_error": err,
            "exact_algorithm": exact,
            "speedup_vs_cpu_all_cores": (value / world / baseline["value"]) if baseline else None,
            "speedup_vs_cpu_single_thread": (value / world / baseline["single_thread_value"]) if baseline else None,
            "reference_counters_sample": ref_counters,
            "nan_pixels": int((pixels == 0x80000000).sum()),
        }
        print(json.dumps(line), file=out, flush=True)
    g.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
