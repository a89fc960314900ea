#!/usr/bin/env python3
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cs397raytracingsp22_amd import Context, scenes, dist as pdist
sc = scenes.config2(1920, 1080, 256, 10)
ctx = Context(0); ctx.upload(sc.flatten())
dev = torch.device("cuda:0")
for world in (1, 8):
    padded = pdist.tiles_padded(1920, 1080, world)
    buf = torch.empty((padded, 1024, 3), dtype=torch.float32, device=dev)
    ctx.reserve(sc.camera, world)
    for rep in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        ctx.render_tiles_device(sc.camera, buf.data_ptr(), None, seed=1, rank=0, world=world)
        torch.cuda.synchronize(); wall = (time.perf_counter() - t0) * 1e3
    p = ctx.last_pipeline_ms()
    print(f"world={world}: wall {wall:.2f} ms  K1w events {ctx.last_kernel_ms():.2f} ms  kernels sum {p['wf_main_ms'] + p['wf_trav_ms'] + p['wf_reduce_ms']:.2f} ms  {p}", flush=True)
