#!/usr/bin/env python3
"""bench.py — the driver's benchmark contract for the path-tracing hot path.

  python bench.py --gpus N --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
         --master-port P bench.py --gpus N --steps K --warmup W

A "step" is one pass of the hot path over one frame: K1w (the wavefront path-tracing
pipeline over this rank's tiles: wf_main / wf_trav once per path segment + wf_reduce) -> the
frame's single gather (RCCL over xGMI, N > 1 only) -> K3 un-permute -> K4 tone-map on rank 0.  Workload = BASELINE.json configs[1]: Cornell box + Utah teapot
BVH, 1920x1080, 256 spp, path_depth 10, synthetic scene built from the reference's
primitive types (cs397raytracingsp22_amd/scenes.py).  The SAME frame is rendered at every
N (tiles sharded over ranks), so scaling is "strong"; `value` = W*H*spp / step time,
whole job, scene and all buffers resident in HBM before the timed region.

Rank 0 prints ONE JSON line with `roofline` and (N = 1) `cpu_baseline` objects.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0      # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def algorithmic_bytes_per_sample(cfg: str, spp: int):
    """B_sample = S*B_list + N_box*32 + N_tri*36 + N_meshhit*60 + N_texel*3 + 12/spp,
    with the per-sample counters measured by the CPU oracle (tests/golden/workcounts_<cfg>.json)."""
    with open(os.path.join(ROOT, "tests", "golden", f"workcounts_{cfg}.json")) as fh:
        wc = json.load(fh)
    p = wc["per_sample"]
    b = (p["segments"] * wc["b_list_bytes"] + p["box_tests"] * 32.0 + p["tri_tests"] * 36.0 +
         p["mesh_hits"] * 60.0 + p["texel_fetches"] * 3.0 + 12.0 / spp)
    return b, p["segments"]


def measured_traffic(cfg: str):
    """HBM bytes per K1 launch from the rocprofv3 PMC passes committed under profiles/
    (FETCH_SIZE and WRITE_SIZE in separate passes, gfx950 read correction applied), or None."""
    path = os.path.join(ROOT, "profiles", f"traffic_{cfg}.json")
    if not os.path.exists(path):
        return None
    with open(path) as fh:
        return json.load(fh).get("hbm_bytes_per_launch")


def cpu_baseline(sc, flat):
    """The plain-C oracle (a port: the Rust reference cannot be built here) timed on this
    host's cores on a bounded sample of the same workload."""
    from oracle import orc_py
    cam = sc.camera
    threads = min(orc_py.usable_cores(), 64)
    o = orc_py.OracleScene(flat)
    W, H = cam.screen_width, cam.screen_height
    # rows are the parallel unit (tracing.rs:228).  Time-boxed: successive sets of `threads` rows,
    # each set spread evenly over the whole image height (same mix of cheap and expensive pixels),
    # until ~12 s of CPU work have been done.
    rows_per_set = max(4 * threads, 32)          # several rows per worker: the dynamic row queue balances them
    stride = max(1, H // rows_per_set)
    rows_per_set = min(rows_per_set, H // stride)
    n, sets, dt = 0, 0, 0.0
    while dt < 12.0 and sets < stride:
        t0 = time.perf_counter()
        o.render(cam, seed=1, threads=threads, window=(0, sets, W, rows_per_set), row_stride=stride,
                 want_u8=False, want_sig=False)
        dt += time.perf_counter() - t0
        n += W * rows_per_set * cam.aa_sample_count
        sets += 1
    rows = rows_per_set * sets
    return {"value": n / dt / 1e6, "unit": "Msamples/s", "cores": threads, "kind": "port",
            "sample": f"{rows} full-width rows ({sets} sets of every {stride}th row) of the same {W}x{H} frame, all {cam.aa_sample_count} spp: "
                      f"{n} samples in {dt:.1f} s (plain-C oracle, one task per scanline like rayon, tracing.rs:228)"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", default="cfg2", choices=["cfg1", "cfg2", "cfg3"])
    ap.add_argument("--spp", type=int, default=0, help="override samples per pixel (debug; invalidates the metric)")
    ap.add_argument("--variant", type=int, default=0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo only to rehearse N>1 on one GPU)")
    ap.add_argument("--same-gpu", action="store_true", help="rehearsal: every rank uses GPU 0 (with --backend gloo)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from cs397raytracingsp22_amd import Context, scenes
    from cs397raytracingsp22_amd.dist import TiledRenderer

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
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

    if args.config == "cfg1":
        sc, cfgname = scenes.config1(), "cfg1"
    elif args.config == "cfg3":
        sc, cfgname = scenes.config3(), "cfg2"
    else:
        sc, cfgname = scenes.config2(), "cfg2"
    if args.spp:
        sc.camera.aa_sample_count = args.spp
    cam = sc.camera
    flat = sc.flatten()
    ctx = Context(local_rank)
    ctx.upload(flat)                      # scene resident in HBM before the timed region
    r = TiledRenderer(ctx, cam, rank=rank, world=world, device=str(device), variant=args.variant)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(device)

    for _ in range(args.warmup):
        r.render_frame(seed=1)
    barrier()
    t0 = time.perf_counter()
    pipe = {"wf_main_ms": 0.0, "wf_trav_ms": 0.0, "wf_reduce_ms": 0.0, "launches": 0}
    for s in range(args.steps):
        r.render_frame(seed=1 + s, time_kernel=True)
        for k, v in ctx.last_pipeline_ms().items():
            pipe[k] += v
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

    samples_per_frame = cam.screen_width * cam.screen_height * cam.aa_sample_count
    ms_per_step = dt / args.steps * 1e3
    value = samples_per_frame * args.steps / dt / 1e6

    if rank == 0:
        b_sample, segs = algorithmic_bytes_per_sample(cfgname, cam.aa_sample_count)
        # dominant kernel = K1w, the wavefront pipeline: one pipeline pass traces this rank's share of
        # the frame (kernel_ms = HIP events around the whole pass; per-kernel sums from events around
        # every launch).  The megakernel variants (--variant 1..6) are a single launch.
        samples_per_launch = samples_per_frame / world
        achieved = b_sample * samples_per_launch / (kernel_ms * 1e-3) / 1e9
        per_step = {k: (v / args.steps) for k, v in pipe.items()}
        traffic = measured_traffic(cfgname) if world == 1 and not args.spp else None
        out = {
            "metric": "Msamples/sec (=rays/sec) at 1080p Cornell+teapot, 256 spp; 1/2/4/8 GPU",
            "value": value, "unit": "Msamples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic", "backend": (args.backend if world > 1 else None),
            "config": {"workload": f"{args.config}: Cornell box (10 Triangle + 2 Sphere) + teapot StaticMesh (240 tris, reference-topology BVH)"
                                   if cfgname == "cfg2" else "cfg1: Cornell box (10 Triangle + 2 Sphere)",
                       "width": cam.screen_width, "height": cam.screen_height, "spp": cam.aa_sample_count,
                       "path_depth": cam.path_depth, "parallelism": f"tiles32x32_mod{world}",
                       "caller": "python ctypes over the C ABI (include/mi_rt.h)",
                       "segments_per_sample": segs, "msegments_per_s": value * segs},
            "roofline": {"bound": "hbm", "kernel": "K1w pipeline (wf_main + wf_trav per segment, wf_reduce)" if args.variant in (0, 7) else "pt_megakernel",
                         "achieved": achieved, "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "kernel_ms": kernel_ms, "algorithmic_bytes_per_sample": b_sample,
                         # per-kernel HIP events cost ~0.4 ms of barriers per frame: the library records them on
                         # single-rank renders only (MI_RT_WF_KERNEL_TIMING=1 forces them)
                         "per_step_ms": ({"wf_main": per_step["wf_main_ms"], "wf_trav": per_step["wf_trav_ms"],
                                          "wf_reduce": per_step["wf_reduce_ms"], "launches": per_step["launches"]}
                                         if per_step["launches"] else None),
                         "note": "algorithmic bytes = what traversal dereferences (SURVEY.md §8d): object list and BVH are "
                                 "SGPR/LDS-resident; `traffic` = PMC-measured HBM bytes per pipeline pass, i.e. the path "
                                 "state streamed between the phase kernels plus the framebuffer"},
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(sc, flat)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    ctx.close()


if __name__ == "__main__":
    main()
