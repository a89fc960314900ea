#!/usr/bin/env python3
"""bench.py — the driver's benchmark contract for the path-tracing hot path.

  python bench.py --gpus N --steps K --warmup W          any N >= 1, no launcher: N > 1 runs IN THIS PROCESS through the
                                                         library's own multi-GPU entry point (mi_multi_*: native RCCL fan-in)
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
         --master-port P bench.py --gpus N --steps K --warmup W        one process per GPU (dist.py, torch.distributed = RCCL)

A "step" is one pass of the hot path over one frame: K1w (the wavefront path-tracing
pipeline over this rank's tiles: wf_main / wf_trav once per path segment + wf_reduce) -> the
frame's single gather (RCCL over xGMI, N > 1 only) -> K3 un-permute -> K4 tone-map on rank 0.
Default workload = BASELINE.json configs[1]: Cornell box + Utah teapot BVH, 1920x1080, 256 spp,
path_depth 10, synthetic scene built from the reference's primitive types
(cs397raytracingsp22_amd/scenes.py); --config selects another BASELINE configuration.  The SAME
frame is rendered at every N (tiles sharded over ranks), so scaling is "strong"; `value` =
W*H*spp / step time, whole job, scene and all buffers resident in HBM before the timed region.

Rank 0 prints ONE JSON line.  `roofline` is the HBM roofline of the pipeline: achieved = HBM bytes per frame /
pipeline time, never the "logical" bytes of SURVEY.md 8(d) (those are served from SGPRs / LDS and
exceed what HBM could deliver; they are printed as `algorithmic_bytes_per_sample`, informational).
HBM bytes come from the rocprofv3 PMC passes committed under profiles/ WHEN they were measured on the
kernel sources of this checkout (source hash), otherwise from the traffic model evaluated on this run's
own path counts (mi_last_pipeline_counts); `traffic_source` says which.  `roofline.valu` is the
second bound of this VALU-heavy path (issue utilisation and live lanes, committed PMC only).
"""
import argparse
import hashlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0      # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (about 6.3 TB/s achievable)
KERNEL_SOURCES = ["cs397raytracingsp22_amd/csrc/pt_kernels.hip", "cs397raytracingsp22_amd/csrc/pt_device.h",
                  "cs397raytracingsp22_amd/csrc/mi_rt.cpp"]

CONFIGS = {   # name: (scene factory, workload text, workcounts file key, BASELINE.json metric config?)
    "cfg1": "configs[0] Cornell box (10 Triangle + 2 Sphere), 400x400, 16 spp, depth 8",
    "cfg2": "configs[1] Cornell box (10 Triangle + 2 Sphere) + teapot StaticMesh (240 tris, reference-topology BVH), 1920x1080, 256 spp",
    "cfg3": "configs[2] same scene, 1920x1080, 1024 spp (the 8-GPU tiling configuration)",
    "cfg4": "configs[3] textured drone StaticMesh (1736 tris, five synthetic 2048^2 maps incl. normal map) in the Cornell box, 1920x1080, 256 spp",
    "cfg5": "configs[4] dielectric sphere enclosing an isotropic ConvexVolume + glass + metal, 1920x1080, 4096 spp, depth 50",
    "head": "the reference's own run() scene (tracing.rs:356-543: drone + cube + 32512-triangle sphere meshes, 15 ParameterizedMaterial "
            "spheres, 2 volumes, plane, light) at the published 800x800, 256 spp",
}


def make_scene(name):
    from cs397raytracingsp22_amd import scenes
    if name == "cfg1":
        return scenes.config1()
    if name == "cfg2":
        return scenes.config2()
    if name == "cfg3":
        return scenes.config3()
    if name == "cfg4":
        return scenes.config4()
    if name == "cfg5":
        return scenes.config5()
    return scenes.head_scene(800, 800, 256, 10, textures=scenes.load_asset_textures())


def source_hash():
    h = hashlib.sha1()
    for f in KERNEL_SOURCES:
        with open(os.path.join(ROOT, f), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:12]


def algorithmic_bytes_per_sample(cfg: str, spp: int):
    """B_sample = S*B_list + N_box*32 + N_tri*36 + N_meshhit*60 + N_texel*3 + 12/spp,
    with the per-sample counters measured by the CPU oracle (tests/golden/workcounts_<cfg>.json).
    LOGICAL bytes (what traversal dereferences, cache-less): informational, never a roofline numerator."""
    path = os.path.join(ROOT, "tests", "golden", f"workcounts_{'cfg2' if cfg == 'cfg3' else cfg}.json")
    if not os.path.exists(path):
        return None, None
    with open(path) as fh:
        wc = json.load(fh)
    p = wc["per_sample"]
    b = (p["segments"] * wc["b_list_bytes"] + p["box_tests"] * 32.0 + p["tri_tests"] * 36.0 +
         p["mesh_hits"] * 60.0 + p["texel_fetches"] * 3.0 + 12.0 / spp)
    return b, p["segments"]


def committed_pmc(cfg: str):
    """profiles/traffic_<cfg>.json (tools/prof.sh + tools/prof_json.py) if it was measured on THESE kernel sources."""
    path = os.path.join(ROOT, "profiles", f"traffic_{cfg}.json")
    if not os.path.exists(path):
        return None, "no committed PMC profile for this configuration"
    with open(path) as fh:
        t = json.load(fh)
    if t.get("source_hash") != source_hash():
        return None, f"committed PMC profile {t.get('tag')} was measured on other kernel sources ({t.get('source_hash')} != {source_hash()})"
    return t, None


def traffic_model(counts, n_meshes):
    """HBM bytes of one pipeline pass from this run's own path counts: every path that
    survives a pass is written once and read once (72 B class A, 76 B class B), a class-B path (the walkers' work list;
    there is no queue) costs the walker's fetch of origin, direction and hit record (40 B), every sample
    slot is written once and read once by wf_reduce (16 B each), and the frame leaves as 16 B accumulator +
    12 B compact pixel.  An upper bound: the samples of dead tiles are neither written nor read."""
    state = 2 * (counts["paths_a"] * 72 + counts["paths_b"] * 76)
    queue = counts["queue_entries"] * (40 + (16 if n_meshes > 1 else 0))
    samp = counts["sample_slots"] * 32
    fb = counts["pixels"] * 28
    return float(state + queue + samp + fb)


def cpu_baseline(sc, flat):
    """The plain-C oracle (a port: the Rust reference cannot be built here) timed on this host's cores on a
    bounded sample of the same workload: built -O3 -march=native on this host, one task
    per scanline like rayon (tracing.rs:228), every usable core."""
    from oracle import orc_py
    cam = sc.camera
    threads = orc_py.usable_cores()
    build = "-O3 -march=native"
    try:
        lib = orc_py.load(orc_py.build_native())
    except Exception as e:      # no compiler on this host: the portable -O2 build, and say so
        lib, build = None, f"-O2 portable (native build failed: {type(e).__name__})"
    o = orc_py.OracleScene(flat, lib=lib)
    W, H = cam.screen_width, cam.screen_height
    # Bounded sample: sets of rows spread evenly over the image height (the same mix of cheap and expensive pixels as
    # the whole frame).  A short calibration set sizes the measured one to about 12 s of wall time on every usable core.
    def run(first_row, n_rows, stride):
        t0 = time.perf_counter()
        o.render(cam, seed=1, threads=threads, window=(0, first_row, W, n_rows), row_stride=stride, want_u8=False, want_sig=False)
        return time.perf_counter() - t0
    cal_rows = min(H, max(threads, 8))
    cal_stride = max(1, H // cal_rows)
    t_cal = run(0, cal_rows, cal_stride)
    rows = int(min(H - cal_rows, max(cal_rows, cal_rows * 12.0 / max(t_cal, 1e-3))))
    rows = max(threads, rows - rows % max(1, threads))
    stride = max(1, H // rows)
    rows = min(rows, H // stride)
    dt = run(min(1, stride - 1), rows, stride)
    n = W * rows * cam.aa_sample_count
    model = cpu_model()
    return {"value": n / dt / 1e6, "unit": "Msamples/s", "cores": threads, "kind": "port", "cpu": model, "build": build,
            "sample": f"{rows} full-width rows (every {stride}th row) of the same {W}x{H} frame, all {cam.aa_sample_count} spp: "
                      f"{n} samples in {dt:.1f} s after a {cal_rows}-row calibration pass (plain-C oracle, one task per scanline "
                      f"like rayon, tracing.rs:228; cores = worker threads = affinity mask capped by the cgroup CPU quota)"}


def cpu_model():
    try:
        with open("/proc/cpuinfo") as fh:
            for line in fh:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown CPU"


def report(args, sc, flat, world, backend, route, dt, kernel_ms, pipe, counts, extra_config=None):
    """Rank 0's ONE JSON line.  `kernel_ms` = pipeline pass of the slowest device (HIP events on its launch stream), `pipe` the
    per-kernel sums and `counts` the path counts of ONE device (device 0) over the timed steps."""
    cam = sc.camera
    samples_per_frame = cam.screen_width * cam.screen_height * cam.aa_sample_count
    ms_per_step = dt / args.steps * 1e3
    value = samples_per_frame * args.steps / dt / 1e6
    b_sample, segs_oracle = algorithmic_bytes_per_sample(args.config, cam.aa_sample_count)
    wavefront = args.variant in (0, 7)
    # What the headline counts.  `value` divides ALL W*H*spp samples by the time, as the metric is defined; the samples of dead
    # tiles (32x32 tiles from which no camera ray can reach anything: black by tracing.rs:306, no ray is generated) cost nothing,
    # so the rate over the pixels that see something is printed beside it, and the segment rate comes from the device's own count of
    # Scene::intersect_ray evaluations of the timed steps — not from the oracle's per-sample mean, which counts one segment for
    # every void sample.  `counts` are device 0's: with N > 1 ranks the segment figures are left out.
    dead_frac = live_rate = segs = mseg = None
    if counts and wavefront and world == 1:
        dead_frac = counts["dead_tile_samples"] / float(samples_per_frame)
        live_rate = (samples_per_frame - counts["dead_tile_samples"]) * args.steps / dt / 1e6
        if counts.get("segments"):
            segs = counts["segments"] / float(samples_per_frame)
            mseg = counts["segments"] * args.steps / dt / 1e6
    per_step = {k: (v / args.steps) for k, v in pipe.items()}
    # ---- HBM roofline of the dominant kernel group (the pipeline pass of ONE device) ----
    pmc, why_not = (committed_pmc(args.config) if world == 1 and not args.spp and wavefront and not args.flags and not args.max_state_gb
                    else (None, "not the profiled configuration"))
    model_bytes = traffic_model(counts, flat.desc.n_meshes) if (wavefront and counts and counts["passes"]) else None
    if pmc is not None and pmc.get("hbm_bytes_per_launch"):
        traffic, traffic_source = float(pmc["hbm_bytes_per_launch"]), \
            f"committed rocprofv3 PMC, profiles/traffic_{args.config}.json tag {pmc['tag']} (same kernel sources {pmc['source_hash']}); not measured in this run"
    elif model_bytes is not None:
        traffic, traffic_source = model_bytes, f"traffic model on this run's own path counts of device 0 ({why_not})"
    else:
        traffic, traffic_source = None, why_not
    achieved = traffic / (kernel_ms * 1e-3) / 1e9 if (traffic and kernel_ms) else None
    frac = achieved / HBM_PEAK_GBS if achieved is not None else None
    # The guide's x2 on FETCH_SIZE is calibrated on wide coalesced reads; the walkers' per-ray gathers are not that, so the
    # corrected figure is an upper bound and the raw counters a lower one: the fraction is a range.
    frac_raw = None
    if pmc is not None and pmc.get("hbm_bytes_raw_per_launch") and kernel_ms:
        frac_raw = float(pmc["hbm_bytes_raw_per_launch"]) / (kernel_ms * 1e-3) / 1e9 / HBM_PEAK_GBS
    if frac is not None:
        assert frac <= 1.0, f"HBM roofline fraction {frac} > 1: traffic accounting is wrong"
    valu = dom = None
    valu_issue = valu_lane = None
    if pmc is not None:
        valu = {k: {"ms": v["ms"], "valu_insts": v["valu_insts"], "issue_frac": v["valu_issue_frac"], "active_lanes": v["active_lanes"],
                    "cycles_per_inst": v.get("valu_cycles_per_inst"), "issue_frac_of_measured_peak": v.get("valu_issue_frac_of_measured_peak"),
                    "lane_frac": (v["valu_issue_frac"] * v["active_lanes"] / 64.0 if v.get("valu_issue_frac") and v.get("active_lanes") else None),
                    "hbm_GBps": v["hbm_GBps"], "hbm_frac": (v["hbm_GBps"] / HBM_PEAK_GBS if v.get("hbm_GBps") else None)}
                for k, v in pmc.get("per_kernel", {}).items() if v.get("valu_insts") or v.get("hbm_bytes")}
        if valu:
            dom = max(valu, key=lambda k: valu[k]["ms"])
            valu_issue, valu_lane = valu[dom]["issue_frac"], valu[dom]["lane_frac"]
    out = {
        "metric": "Msamples/sec (=rays/sec) at 1080p Cornell+teapot, 256 spp; 1/2/4/8 GPU",
        "value": value, "unit": "Msamples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic", "backend": backend,
        "config": {"workload": f"{args.config}: {CONFIGS[args.config]}",
                   "width": cam.screen_width, "height": cam.screen_height, "spp": cam.aa_sample_count,
                   "path_depth": cam.path_depth, "parallelism": f"tiles32x32_mod{world}",
                   "max_state_gb": (args.max_state_gb or "library default"),
                   "caller": "python ctypes over the C ABI (include/mi_rt.h)", "route": route,
                   "dead_tile_frac": dead_frac, "live_pixel_msamples_per_s": live_rate,
                   "segments_per_sample": segs, "msegments_per_s": mseg, "segments_source": "device counter (wf_main), last timed step",
                   "segments_per_sample_oracle": segs_oracle,
                   "multi_gpu_note": "N>1 numbers exist only where this script ran on a multi-GPU node; "
                                     "the builder's own N>1 figures are single-GPU rehearsals (DESIGN.md section 6)"},
        # Two resources bound this path and NEITHER is saturated: `frac` is the HBM fraction of the
        # pipeline pass (the key the bench contract names), `valu_lane_frac` the VALU lane-throughput fraction of the
        # dominant kernel.  "bound" names the roofline `achieved`/`peak`/`frac` are quoted on, not a claim that it binds.
        "roofline": {"bound": "hbm",
                     "kernel": "K1w pipeline pass (wf_main + wf_prefix + wf_trav per segment, wf_reduce)" if wavefront else "single-launch kernel",
                     "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": frac,
                     "frac_x2_corrected": (frac if pmc is not None else None), "frac_raw_counters": frac_raw,
                     "valu_issue_frac": valu_issue, "valu_lane_frac": valu_lane, "valu_kernel": dom,
                     "binding": "VALU instruction issue on partly filled waves. Measured ceiling on gfx950 (tools/microbench/valu_rates.hip): one wave64 "
                                "VALU instruction per ~2.5 cycles per SIMD; the walkers run at 2.4-2.6 cycles per instruction and wf_main's class-A form at "
                                "2.7 (valu.*.cycles_per_inst, from SQ_INSTS_VALU and GRBM_GUI_ACTIVE of the committed PMC) with 37-52 of 64 lanes live, so "
                                "kernel time follows the instruction count (walkers: -4.6 % instructions = -4.4 % time, paired-layout A/B); wf_main's "
                                "class-B form adds the latency of its per-lane gathers (3.4 cycles per instruction); HBM carries the path state at "
                                "0.21-0.28 of its peak. valu_issue_frac keeps the nominal 2 cycles / 2.4 GHz of earlier rounds for comparison",
                     "traffic": traffic, "traffic_source": traffic_source,
                     "traffic_model_bytes": model_bytes, "path_counts": counts,
                     "kernel_ms": kernel_ms,
                     "per_step_ms": ({"wf_main": per_step["wf_main_ms"], "wf_trav": per_step["wf_trav_ms"],
                                      "wf_reduce": per_step["wf_reduce_ms"], "launches": per_step["launches"],
                                      "wf_trav_f": per_step["wf_trav_f_ms"], "wf_replay": per_step["wf_replay_ms"],
                                      "wf_main_class_a_beside_walkers": per_step["wf_main_a_ms"]}
                                     if per_step.get("launches") else None),
                     "valu": valu,
                     "algorithmic_bytes_per_sample": b_sample,
                     "note": "achieved = HBM bytes of one device's pipeline pass / its duration (HIP events on the launch stream). "
                             "The path state streamed between the phase kernels IS the traffic; the scene (object list, BVH) is "
                             "SGPR/LDS/L2-resident, so SURVEY 8(d)'s logical bytes are informational only. valu_issue_frac = "
                             "SQ_INSTS_VALU x 2 cycles / (1024 SIMDs x 2.4 GHz x kernel time); valu_lane_frac = valu_issue_frac x "
                             "active lanes / 64 (committed PMC of the same kernel sources only)"},
    }
    if extra_config:
        out["config"].update(extra_config)
    if world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(sc, flat)
    print(json.dumps(out), flush=True)


def run_in_process(args):
    """N GPUs of this node from ONE process, through the product's own multi-GPU entry point (mi_multi_*: one context,
    stream and host thread per device inside the library, native RCCL send/recv fan-in to device 0, K3 + K4 there) —
    what a Rust caller's `render_to_image` (tracing.rs:221) would call.  No launcher, no torch: every output pointer is
    NULL, so the finished f32 and u8 images stay resident on device 0 and nothing crosses PCIe in the timed region."""
    from cs397raytracingsp22_amd import MultiContext
    sc = make_scene(args.config)
    if args.spp:
        sc.camera.aa_sample_count = args.spp
    cam = sc.camera
    flat = sc.flatten()
    # --loopback: the TEST transport (mi_multi_create_loopback) — every rank on device 0, event-ordered device-to-device copies instead of
    # the RCCL pairs.  It runs mi_multi_render's N >= 2 code end to end on a one-GPU box; the ranks share the card, so `value` says what
    # the multi-rank machinery costs on top of one rank, NOT how it scales.
    m = MultiContext.loopback(args.gpus) if args.loopback else MultiContext(args.gpus)            # fails loudly (MI_ERR_INVALID: device out of range) when the node has fewer devices
    try:
        m.upload(flat)                     # scene resident in HBM (replicated) before the timed region
        msb = int(args.max_state_gb * 1e9)
        m.reserve(cam, msb)
        dev0 = m.context(0)
        for _ in range(args.warmup):
            m.render(cam, seed=1, want_f32=False, want_u8=False, variant=args.variant, flags=args.flags, max_state_bytes=msb)
        pipe = {"wf_main_ms": 0.0, "wf_trav_ms": 0.0, "wf_reduce_ms": 0.0, "launches": 0, "wf_trav_f_ms": 0.0, "wf_replay_ms": 0.0, "wf_main_a_ms": 0.0}
        counts, kms, wall = None, [], []
        # mi_multi_render is blocking: it returns after every device's stream has drained (the barrier + synchronize of the
        # contract are inside the call, on both sides of every step)
        t0 = time.perf_counter()
        for s in range(args.steps):
            _, _, _, st = m.render(cam, seed=1 + s, want_f32=False, want_u8=False, variant=args.variant, flags=args.flags, max_state_bytes=msb)
            kms.append(st.kernel_ms)       # the slowest device's pipeline pass
            wall.append(st.total_ms)
            for k, v in dev0.last_pipeline_ms().items():
                pipe[k] += v
            counts = dev0.last_pipeline_counts()
        dt = time.perf_counter() - t0
        report(args, sc, flat, args.gpus, "loopback test transport (all ranks on device 0: NOT a scaling figure)" if args.loopback else "rccl (mi_multi, in-process)",
               "mi_multi_* in one process (loopback test transport)" if args.loopback else "mi_multi_* in one process (native RCCL fan-in, include/mi_rt.h)",
               dt, sum(kms) / max(1, len(kms)), pipe, counts,
               extra_config={"ranks": m.n_devices, "mi_multi_total_ms": sum(wall) / max(1, len(wall))})
    finally:
        m.close()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", default="cfg2", choices=sorted(CONFIGS))
    ap.add_argument("--spp", type=int, default=0, help="override samples per pixel (debug; invalidates the metric)")
    ap.add_argument("--variant", type=int, default=0)
    ap.add_argument("--flags", type=int, default=0, help="mi_render_opts.flags (MI_OPT_*), developer A/B")
    ap.add_argument("--max-state-gb", type=float, default=0.0, help="mi_render_opts.max_state_bytes in GB (0 = the library's default budget)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--loopback", action="store_true", help="N ranks on device 0 through mi_multi_create_loopback (test transport; exercises the N >= 2 code, measures no scaling)")
    ap.add_argument("--via-multi", action="store_true", help="N = 1 through mi_multi_* as well (the route every N > 1 run without a launcher takes)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo only to rehearse N>1 on one GPU)")
    ap.add_argument("--same-gpu", action="store_true", help="rehearsal: every rank uses GPU 0 (with --backend gloo)")
    args = ap.parse_args()
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")

    # Two ways to N GPUs.  Under torch.distributed.run (WORLD_SIZE set): one process per GPU, dist.py + torch's RCCL gather.
    # Without a launcher: N > 1 (or --via-multi) runs in THIS process through mi_multi_* (native RCCL inside the library).
    launched = "WORLD_SIZE" in os.environ
    if not launched and (args.gpus > 1 or args.via_multi or args.loopback):
        return run_in_process(args)

    import torch
    import torch.distributed as dist
    from cs397raytracingsp22_amd import Context
    from cs397raytracingsp22_amd.dist import TiledRenderer

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but the launcher set WORLD_SIZE={world}: pass --nproc-per-node {args.gpus}")
    if args.same_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    device = torch.device(f"cuda:{local_rank}")
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=device)
        else:
            dist.init_process_group(backend=args.backend)

    sc = make_scene(args.config)
    if args.spp:
        sc.camera.aa_sample_count = args.spp
    cam = sc.camera
    flat = sc.flatten()
    ctx = Context(local_rank)
    ctx.upload(flat)                      # scene resident in HBM before the timed region
    r = TiledRenderer(ctx, cam, rank=rank, world=world, device=str(device), variant=args.variant, flags=args.flags,
                      max_state_bytes=int(args.max_state_gb * 1e9))

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(device)

    for _ in range(args.warmup):
        r.render_frame(seed=1)
    barrier()
    t0 = time.perf_counter()
    pipe = {"wf_main_ms": 0.0, "wf_trav_ms": 0.0, "wf_reduce_ms": 0.0, "launches": 0, "wf_trav_f_ms": 0.0, "wf_replay_ms": 0.0, "wf_main_a_ms": 0.0}
    counts = None
    for s in range(args.steps):
        r.render_frame(seed=1 + s, time_kernel=True)
        for k, v in ctx.last_pipeline_ms().items():
            pipe[k] += v
        counts = ctx.last_pipeline_counts()
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        km = torch.tensor([sum(r.kernel_ms) / max(1, len(r.kernel_ms))], dtype=torch.float64, device=device)
        dist.all_reduce(km, op=dist.ReduceOp.MAX)
        kernel_ms = float(km.item())
    else:
        kernel_ms = sum(r.kernel_ms) / max(1, len(r.kernel_ms))

    if rank == 0:
        report(args, sc, flat, world, (args.backend if world > 1 else None),
               "one process per GPU (torch.distributed.run), dist.py + torch.distributed.gather" if world > 1 else "one process, one GPU (mi_ctx_*)",
               dt, kernel_ms, pipe, counts)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    ctx.close()


if __name__ == "__main__":
    main()
