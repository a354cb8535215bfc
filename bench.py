#!/usr/bin/env python3
"""bench.py - Mpaths/s of the radiance path on scenes/cornell.txt 1024x768 (BASELINE.json config 2).

    python bench.py --gpus N --steps K --warmup W

A step = `--passes` progressive passes (one path per pixel per pass) over the whole frame, issued
through the C-ABI as ONE render call (the library splits it into launches: per pass and image region
for this workload).  Inputs (scene records, seeds, colour buffer) are resident
in HBM before the timed region; the region is K steps bracketed by a barrier + device sync, MAX over
ranks.  With N > 1 each rank renders from its own seed block and the colour buffers are summed with
one RCCL all-reduce inside the timed region (weak scaling: per-GPU work is fixed).

The JSON line also carries
  roofline     FP32 VALU: algorithmic flops F per path x paths per launch / average
               launch duration from HIP events on the library's stream, against 157.3 TFLOP/s.  F is
               counted by the CPU oracle on a sample of the same scene while it replays the product's
               shadow-ray candidate sets (the tests the algorithm the kernel runs has to make);
               the reference's brute-force count is reported beside it as reference_equivalent_*
               and is NOT what `achieved` is built from.
  cpu_baseline the CPU oracle (a bit-exact port of the reference's main-cpu.c path) timed on this
               host's cores on a bounded sample of the same scene.
"""
import argparse
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(REPO, "path-tracer_amd"))

PEAK_FP32_VALU_TFLOPS = 157.3  # MI355X_MICROARCH.md: 256 CU x 128 lanes x 2 x 2.4 GHz


def cpu_baseline(scene_name, cull_mode):
    """Rank 0, N = 1 only.  oracle/ is used here as the timed CPU baseline and the flop counter."""
    sys.path.insert(0, os.path.join(REPO, "oracle"))
    import oracle_binding as ob
    import pyptgpu as pt
    threads = int(os.environ.get("OMP_NUM_THREADS", "0")) or min(os.cpu_count() or 1, 16)
    # flops per path of the reference algorithm, counted on a 96x72 frame (serial, counting build)
    cw, ch = 96, 72
    sc = pt.LoadedScene(scene_name, cw, ch)
    ctr = ob.Counters()
    t0 = time.perf_counter()
    ob.Render(sc, pt.seed_block(cw, ch)).pass_oracle_counted(ctr, 1)
    t_count = time.perf_counter() - t0
    flops_per_path = ctr.flops() / ctr.paths
    # the same count with the shadow-ray candidate sets the kernel uses (same image under SAFE)
    culled = ob.Counters()
    with ob.shadow_candidates(pt.shadow_candidates(sc, cull_mode)):
        ob.Render(sc, pt.seed_block(cw, ch)).pass_oracle_counted(culled, 1)
    culled_flops_per_path = culled.flops() / culled.paths
    # single thread = the reference's serial loop (main-cpu.c:61-62)
    t0 = time.perf_counter()
    ob.Render(sc, pt.seed_block(cw, ch)).pass_oracle(1, threads=1)
    t_one = time.perf_counter() - t0
    one = cw * ch / t_one / 1e6
    # all threads, sized for ~10-20 s of wall time
    paths_target = max(one * 1e6 * threads * 12.0, cw * ch)
    scale = max(1, int((paths_target / (cw * ch)) ** 0.5))
    bw, bh = min(1024, cw * scale) // 8 * 8, min(768, ch * scale) // 8 * 8
    scb = pt.LoadedScene(scene_name, bw, bh)
    t0 = time.perf_counter()
    ob.Render(scb, pt.seed_block(bw, bh)).pass_oracle(1, threads=threads)
    t_all = time.perf_counter() - t0
    return {
        "value": bw * bh / t_all / 1e6, "unit": "Mpaths/s", "cores": threads, "kind": "port",
        "sample": f"{scene_name} {bw}x{bh} x 1 spp, {threads} OpenMP threads over rows, {t_all:.1f} s",
        "single_thread_value": one,
        "single_thread_sample": f"{scene_name} {cw}x{ch} x 1 spp, serial loop as main-cpu.c, {t_one:.1f} s",
    }, culled_flops_per_path, flops_per_path, {"reference": ctr.as_dict(), "with_candidate_sets": culled.as_dict()}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=12)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--passes", type=int, default=32,
                    help="progressive passes per step (the reference's frame after warm-up is 0.5 s of passes, "
                         "main-gpu.c:774-794: hundreds at this rate)")
    ap.add_argument("--scene", default="cornell")
    ap.add_argument("--width", type=int, default=1024)
    ap.add_argument("--height", type=int, default=768)
    ap.add_argument("--kernel", default="wavefront", choices=["mega", "wavefront"])
    ap.add_argument("--math", default="fast", choices=["fast", "strict"])
    ap.add_argument("--integrator", default="pt", choices=["pt", "dl"])
    ap.add_argument("--cull", default="auto", choices=["off", "safe", "tight", "enclosed", "ball", "auto"],
                    help="shadow-ray candidate sets (auto = enclosed with fast math, safe with strict; ball is opt-in)")
    ap.add_argument("--streams", type=int, default=4, help="image regions on concurrent HIP streams (1..4; without the work queue)")
    ap.add_argument("--work-queue", default="auto", choices=["auto", "on", "off"],
                    help="wavefront kernel: one launch per step whose waves take (pass, tile) units (PTGPU_OPT_WORK_QUEUE)")
    ap.add_argument("--passes-per-launch", type=int, default=0, help="0 = automatic")
    ap.add_argument("--backend", default="nccl", help="process-group backend for N > 1 (nccl = RCCL; gloo to rehearse on one GPU)")
    ap.add_argument("--no-cpu", action="store_true", help="skip the CPU baseline leg")
    args = ap.parse_args()

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
    # The library's four region streams are created BEFORE the process group: HIP spreads streams
    # over GPU_MAX_HW_QUEUES (4) hardware queues in creation order, and with RCCL's streams in
    # first the regions double up on a queue and a pass runs as slowly as with one stream (+14 %).
    w, h = args.width, args.height
    sc = pt.LoadedScene(args.scene, w, h)
    seeds = pt.seed_block(w, h, rank)
    g = pt.Gpu(local_rank, pt.INTEGRATOR_PT if args.integrator == "pt" else pt.INTEGRATOR_DL,
               kernel=pt.KERNEL_WAVEFRONT if args.kernel == "wavefront" else pt.KERNEL_MEGA,
               math=pt.MATH_FAST if args.math == "fast" else pt.MATH_STRICT, streams=args.streams,
               work_queue={"auto": 2, "on": 1, "off": 0}[args.work_queue],
               passes_per_launch=args.passes_per_launch,
               shadow_cull={"off": pt.CULL_OFF, "safe": pt.CULL_SAFE, "tight": pt.CULL_TIGHT, "enclosed": pt.CULL_ENCLOSED,
                            "ball": pt.CULL_BALL, "auto": pt.CULL_AUTO}[args.cull])
    cull = args.cull if args.cull != "auto" else ("enclosed" if args.math == "fast" else "safe")
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
    if dist is not None:
        t = torch.tensor([elapsed], device="cuda", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    paths_per_launch = w * h * args.passes
    total_paths = paths_per_launch * args.steps * world
    value = total_paths / elapsed / 1e6

    if rank == 0:
        pixels = g.pixels()
        baseline, flops_per_path, ref_flops_per_path, counters = (None, None, None, None)
        if world == 1 and not args.no_cpu:
            baseline, flops_per_path, ref_flops_per_path, counters = cpu_baseline(
                args.scene, {"off": pt.CULL_OFF, "safe": pt.CULL_SAFE, "tight": pt.CULL_TIGHT, "enclosed": pt.CULL_ENCLOSED,
                             "ball": pt.CULL_BALL}[cull])
        # One step = args.passes passes.  With the work queue a step is ONE launch (rocprof's AverageNs
        # = avg_step_event_ms).  Without it, with S region streams, a pass is S concurrent launches of
        # the same kernel (each 1/S of the tiles); the HIP events bracket the whole step on the
        # library's stream (fork/join), so the per-launch figure below is step time / sequential
        # launch groups (see DESIGN.md "Measurement").
        avg_step_ms = sum(launch_ms) / len(launch_ms)
        n_tiles = ((w + 7) // 8) * ((h + 7) // 8)
        info = g.scene_info()
        # the library's own rule (ptgpu.h, PTGPU_OPT_WORK_QUEUE): one launch per step when it applies
        work = info["primitives"] * (1 + info["lights"])
        unit_passes = min(max((4096 + work - 1) // max(work, 1), 1), 16, args.passes)
        queue = args.kernel == "wavefront" and (args.work_queue == "on" or (
            args.work_queue == "auto" and args.integrator == "pt" and n_tiles >= 2048 and unit_passes * work >= 512))
        ppl = args.passes if queue else (args.passes_per_launch or (1 if (args.streams > 1 and n_tiles >= 2048) else 8))
        groups_per_step = (args.passes + ppl - 1) // ppl
        # HBM bytes per pass from the committed PMC run of this workload (separate rocprofv3 --pmc
        # passes, profiles/traffic.json); null for any other workload.
        traffic = None
        try:
            tj = json.load(open(os.path.join(REPO, "profiles", "traffic.json")))
            if (args.scene, w, h, args.kernel, args.math, cull) == ("cornell", 1024, 768, "wavefront", "fast", tj.get("cull", "off")):
                traffic = tj["bytes_per_pass"] * args.passes
        except (OSError, ValueError, KeyError):
            pass
        roofline = None
        if flops_per_path is not None:
            # primitive tests of the extension rays under a full scan: R x (20 per sphere + 52 per triangle)
            ref_ctr = counters["reference"]
            o = sc.objects
            n_sph, n_tri = int((o["type"] == 0).sum()), int((o["type"] == 1).sum())
            ext_test_flops_per_path = ref_ctr["extension_rays"] * (20 * n_sph + 52 * n_tri) / ref_ctr["paths"]
            achieved = flops_per_path * paths_per_launch / (avg_step_ms * 1e-3) / 1e12
            roofline = {"bound": "valu_fp32", "achieved": achieved, "peak": PEAK_FP32_VALU_TFLOPS,
                        "unit": "TFLOP/s", "frac": achieved / PEAK_FP32_VALU_TFLOPS, "traffic": traffic,
                        "traffic_unit": "HBM bytes per step (PMC FETCH_SIZE + WRITE_SIZE), algorithmic %d" % (52 * w * h * args.passes),
                        "kernel": f"radiance_{args.kernel}_kernel ({args.math})",
                        "avg_step_event_ms": avg_step_ms,
                        "avg_launch_group_ms": avg_step_ms / groups_per_step,
                        "launch_groups_per_step": groups_per_step,
                        "concurrent_launches_per_group": 1 if queue else (args.streams if ppl == 1 or args.streams > 1 else 1),
                        "work_queue": bool(queue),
                        "passes_per_launch": ppl,
                        "algorithmic_flops_per_path": flops_per_path,
                        "algorithm": f"reference integrator with shadow-ray candidate sets ({cull}); extension rays counted "
                                     "as full scans although the fast build walks a hierarchy for them (fewer tests): "
                                     "frac_without_extension_tests is the same figure with those tests left out",
                        "frac_without_extension_tests": (flops_per_path - ext_test_flops_per_path) * paths_per_launch
                                                        / (avg_step_ms * 1e-3) / 1e12 / PEAK_FP32_VALU_TFLOPS,
                        "reference_equivalent_flops_per_path": ref_flops_per_path,
                        "reference_equivalent_tflops": ref_flops_per_path * paths_per_launch / (avg_step_ms * 1e-3) / 1e12,
                        "paths_per_step": paths_per_launch,
                        "hbm_algorithmic_bytes_per_pass": 52 * w * h}
        line = {
            "metric": "Mpaths/s", "value": value, "unit": "Mpaths/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
            "data": "scene file data/scenes/%s.txt (the reference's own), rand()-compatible seeds" % args.scene,
            "config": {"workload": f"{args.scene}.txt {w}x{h}, {args.passes} passes/step, "
                                   f"{args.integrator} integrator, {args.kernel} kernel, {args.math} math, shadow cull {cull}",
                       "spp_timed": args.steps * args.passes, "spp_per_s": args.steps * args.passes * world / elapsed,
                       "parallelism": f"passes split over {world} GPU(s) by seed block + 1 all-reduce"},
            "roofline": roofline, "cpu_baseline": baseline,
            "speedup_vs_cpu_all_cores": (value / world / baseline["value"]) if baseline else None,
            "speedup_vs_cpu_single_thread": (value / world / baseline["single_thread_value"]) if baseline else None,
            "oracle_counters_sample": counters,
            "nan_pixels": int((pixels == 0x80000000).sum()),
        }
        print(json.dumps(line), flush=True)
    g.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
