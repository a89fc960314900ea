import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch, numpy as np
from cs397raytracingsp22_amd import Context, scenes, dist as pdist
sc = scenes.config2(); cam = sc.camera
ctx = Context(0); ctx.upload(sc.flatten())
for world in (8, 4):
    padded = pdist.tiles_padded(cam.screen_width, cam.screen_height, world)
    buf = torch.empty((padded, 1024, 3), dtype=torch.float32, device="cuda:0")
    ctx.reserve(cam, world, 0)
    walls = []
    for r in range(world):
        ctx.render_tiles_device(cam, buf.data_ptr(), None, seed=1, rank=r, world=world)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(5):
            ctx.render_tiles_device(cam, buf.data_ptr(), None, seed=1, rank=r, world=world)
        torch.cuda.synchronize(); walls.append((time.perf_counter() - t0) * 1e3 / 5)
    print("world", world, [round(w, 2) for w in walls], "max/mean", round(max(walls) / np.mean(walls), 3))
