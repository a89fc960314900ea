#!/usr/bin/env python3
"""Predict multi-GPU balance on one GPU: time every rank's tile share of the 1080p/256 spp frame."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cs397raytracingsp22_amd import Context, scenes, dist as pdist
sc = scenes.config2(1920, 1080, 256, 10)
ctx = Context(0); ctx.upload(sc.flatten())
dev = torch.device("cuda:0")
for world in (1, 2, 4, 8):
    padded = pdist.tiles_padded(1920, 1080, world)
    buf = torch.empty((padded, 1024, 3), dtype=torch.float32, device=dev)
    ctx.reserve(sc.camera, world)
    times = []
    for r in range(world):
        for rep in range(2):
            ctx.render_tiles_device(sc.camera, buf.data_ptr(), None, seed=1, rank=r, world=world)
            ms = ctx.last_kernel_ms()
        times.append(ms)
    mx, mean = max(times), sum(times) / len(times)
    print(f"world={world}: per-rank K1w ms min {min(times):.2f} mean {mean:.2f} max {mx:.2f}  balance {mean / mx:.3f}  "
          f"ideal-speedup {times and (sum(times) / mx):.2f}  vs 1-GPU frame: {ref / mx if world > 1 else 1:.2f}x" if world > 1 or not times else f"world=1: {mx:.2f} ms", flush=True)
    if world == 1:
        ref = mx
