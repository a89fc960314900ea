#!/usr/bin/env python3
"""Per-rank WALL time of the frame pipeline on one GPU (kernel events vs wall clock incl. the host loop):
what each of N ranks would spend per frame, without the gather."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cs397raytracingsp22_amd import Context, scenes, dist as pdist
sc = scenes.config2(1920, 1080, 256, 10)
ctx = Context(0); ctx.upload(sc.flatten())
dev = torch.device("cuda:0")
ref = None
for world in (1, 2, 4, 8):
    padded = pdist.tiles_padded(1920, 1080, world)
    buf = torch.empty((padded, 1024, 3), dtype=torch.float32, device=dev)
    ctx.reserve(sc.camera, world)
    walls, kerns = [], []
    for r in range(world):
        ctx.render_tiles_device(sc.camera, buf.data_ptr(), None, seed=1, rank=r, world=world)   # warm
        torch.cuda.synchronize()
        t0 = time.perf_counter(); k = 0.0
        for rep in range(3):
            ctx.render_tiles_device(sc.camera, buf.data_ptr(), None, seed=1, rank=r, world=world)
            k += ctx.last_kernel_ms()
        torch.cuda.synchronize()
        walls.append((time.perf_counter() - t0) * 1e3 / 3); kerns.append(k / 3)
    if world == 1: ref = walls[0]
    pm = ctx.last_pipeline_ms()
    print(f"RES world={world}: last rank pipeline {({k: round(v, 2) for k, v in pm.items()})}")
    print(f"RES world={world}: wall max {max(walls):.2f} mean {sum(walls)/len(walls):.2f} | kernel max {max(kerns):.2f} | host overhead {max(walls)-max(kerns):.2f} ms | speedup vs 1 rank {ref/max(walls):.2f}x", flush=True)
