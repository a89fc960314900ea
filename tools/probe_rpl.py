#!/usr/bin/env python3
"""wf_trav at 1/8 of the frame (rank 3 of 8): per-kernel sums for the current env knobs."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cs397raytracingsp22_amd import Context, scenes, dist as pdist
sc = scenes.config2(1920, 1080, 256, 10)
ctx = Context(0); ctx.upload(sc.flatten())
world = int(os.environ.get("PROBE_WORLD", "8"))
padded = pdist.tiles_padded(1920, 1080, world)
buf = torch.empty((padded, 1024, 3), dtype=torch.float32, device="cuda:0")
ctx.reserve(sc.camera, world)
best = None
for rep in range(4):
    ctx.render_tiles_device(sc.camera, buf.data_ptr(), None, seed=1, rank=3 % world, world=world)
    k = ctx.last_kernel_ms(); pm = ctx.last_pipeline_ms()
    if best is None or k < best[0]: best = (k, pm)
print("RES", os.environ.get("TAG", ""), "frame %.2f ms" % best[0], {a: round(b, 2) for a, b in best[1].items()})
