#!/usr/bin/env python3
"""One parametrised probe of the render path on the GPU box (replaces round 1's one-off probe scripts).

  python tools/probe.py --config cfg2 [--width W --height H --spp N] [--variant V] [--flags F]
                        [--world N] [--reps R] [--max-state-gb G] [--check]

Prints one "RES ..." line per measurement: pipeline pass (HIP events), per-kernel sums, live path counts.
--world N renders every rank's tile share in turn on this one GPU (a rehearsal of the partition, no gather).
--check compares signatures with MI_OPT_REFERENCE_WALK / MI_OPT_NO_TILE_MASKS renders (bit for bit).
"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402,F401
import numpy as np  # noqa: E402
from cs397raytracingsp22_amd import Context, abi, scenes, dist as pdist  # noqa: E402


def make_scene(name, w, h, spp, tex):
    full = {"cfg1": (1920, 1080, 256), "cfg2": (1920, 1080, 256), "cfg3": (1920, 1080, 1024), "cfg4": (1920, 1080, 256),
            "cfg5": (1920, 1080, 4096), "head": (800, 800, 256)}[name]
    w, h, spp = w or full[0], h or full[1], spp or full[2]
    if name == "cfg1":
        return scenes.config1(w, h, spp, 8)
    if name in ("cfg2", "cfg3"):
        return scenes.config2(w, h, spp, 10)
    if name == "cfg4":
        return scenes.config4(w, h, spp, 10, tex_size=tex)
    if name == "cfg5":
        return scenes.config5(w, h, spp, 50)
    return scenes.head_scene(w, h, spp, 10, textures=scenes.load_asset_textures())


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="cfg2")
    ap.add_argument("--width", type=int, default=0)
    ap.add_argument("--height", type=int, default=0)
    ap.add_argument("--spp", type=int, default=0)
    ap.add_argument("--tex", type=int, default=2048)
    ap.add_argument("--variant", type=int, default=0)
    ap.add_argument("--flags", type=int, default=0)
    ap.add_argument("--world", type=int, default=1)
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--max-state-gb", type=float, default=0.0)
    ap.add_argument("--check", action="store_true")
    a = ap.parse_args()
    sc = make_scene(a.config, a.width, a.height, a.spp, a.tex)
    cam = sc.camera
    ctx = Context(0)
    ctx.upload(sc.flatten())
    msb = int(a.max_state_gb * (1 << 30))
    tag = f"{a.config} {cam.screen_width}x{cam.screen_height}x{cam.aa_sample_count} v{a.variant} f{a.flags}"
    if a.world == 1:
        ctx.reserve(cam, 1, msb)
        best = None
        for _ in range(a.reps):
            _, _, _, st = ctx.render(cam, want_u8=False, variant=a.variant, flags=a.flags, max_state_bytes=msb)
            if best is None or st.kernel_ms < best[0]:
                best = (st.kernel_ms, ctx.last_pipeline_ms(), ctx.last_pipeline_counts(), st.samples)
        print(f"RES {tag}: {best[0]:.2f} ms, {best[3] / best[0] / 1e3:.0f} Msamples/s",
              {k: round(v, 2) for k, v in best[1].items()}, best[2], flush=True)
        if os.environ.get("MI_RT_WF_STAMPS"):
            out = (__import__("ctypes").c_uint64 * 16)()
            abi.check(ctx._lib.mi_last_diag(ctx._h, out))
            print("RES diag wf_trav [trips, have, burst steps, burst lanes, leaf steps, leaf lanes, refills, rays]", [int(v) for v in out][:8],
                  "wf_trav_f [trips, walks, burst steps, burst lanes, leaf steps, leaf lanes, refills, entries]", [int(v) for v in out][8:], flush=True)
        if a.check:
            _, _, s0, _ = ctx.render(cam, want_u8=False, want_sig=True, variant=a.variant, flags=a.flags)
            for name, fl in (("reference-walk", abi.MI_OPT_REFERENCE_WALK), ("no-tile-masks", abi.MI_OPT_NO_TILE_MASKS),
                             ("two-stage", abi.MI_OPT_TWO_STAGE)):
                _, _, s1, _ = ctx.render(cam, want_u8=False, want_sig=True, variant=a.variant, flags=fl)
                print(f"RES check {name}: {int((s0 != s1).sum())} of {s0.size} signatures differ", flush=True)
    else:
        dev = torch.device("cuda:0")
        padded = pdist.tiles_padded(cam.screen_width, cam.screen_height, a.world)
        buf = torch.empty((padded, 1024, 3), dtype=torch.float32, device=dev)
        ctx.reserve(cam, a.world, msb)
        walls, kerns = [], []
        for r in range(a.world):
            ctx.render_tiles_device(cam, buf.data_ptr(), None, seed=1, rank=r, world=a.world, flags=a.flags)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            k = 0.0
            for _ in range(a.reps):
                ctx.render_tiles_device(cam, buf.data_ptr(), None, seed=1, rank=r, world=a.world, flags=a.flags)
                k += ctx.last_kernel_ms()
            torch.cuda.synchronize()
            walls.append((time.perf_counter() - t0) * 1e3 / a.reps)
            kerns.append(k / a.reps)
            if os.environ.get("MI_RT_WF_KERNEL_TIMING") == "1" and r in (0, a.world - 1):      # per-kernel sums of the last frame of this rank
                print(f"RES {tag} world={a.world} rank {r}: wall {walls[-1]:.2f} ms", {k2: round(v, 2) for k2, v in ctx.last_pipeline_ms().items()},
                      ctx.last_pipeline_counts(), flush=True)
        print(f"RES {tag} world={a.world}: wall max {max(walls):.2f} mean {np.mean(walls):.2f} ms | kernel max {max(kerns):.2f} ms", flush=True)
    ctx.close()


if __name__ == "__main__":
    main()
