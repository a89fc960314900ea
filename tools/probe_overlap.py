#!/usr/bin/env python3
"""Feasibility probe: do two wavefront pipelines on two streams overlap usefully?"""
import sys, os, time, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cs397raytracingsp22_amd import Context, scenes
sc = scenes.config2(1920, 1080, 128, 10)
flat = sc.flatten()
ctxs = [Context(0), Context(0)]
for c in ctxs:
    c.upload(flat)
    c.render(sc.camera, seed=1, want_u8=False)       # warm-up / allocation
def run(c, seed):
    c.render(sc.camera, seed=seed, want_u8=False, want_f32=False)
for rep in range(3):
    torch.cuda.synchronize()
    t0 = time.perf_counter(); run(ctxs[0], 1); run(ctxs[1], 2); torch.cuda.synchronize(); t_seq = time.perf_counter() - t0
    t0 = time.perf_counter()
    th = [threading.Thread(target=run, args=(ctxs[i], 1 + i)) for i in range(2)]
    [t.start() for t in th]; [t.join() for t in th]; torch.cuda.synchronize(); t_par = time.perf_counter() - t0
    print(f"sequential {t_seq*1e3:.1f} ms   concurrent {t_par*1e3:.1f} ms   ratio {t_seq/t_par:.2f}", flush=True)
