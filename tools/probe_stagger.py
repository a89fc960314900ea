#!/usr/bin/env python3
"""Feasibility probe: what would STAGGERED overlap of two half-frame batches buy?  Two contexts on one device, each renders cfg2 at
half the samples; thread B starts `delay` ms after thread A.  Prints the wall time of the pair against two renders in a row.
(Timing only: the two halves are not accumulated into one image here.)"""
import os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa
from cs397raytracingsp22_amd import Context, scenes

cfg = sys.argv[1] if len(sys.argv) > 1 else "cfg2"
spp_half = int(sys.argv[2]) if len(sys.argv) > 2 else 128
msb = int(float(sys.argv[3]) * (1 << 30)) if len(sys.argv) > 3 else 0          # per-context state budget in GiB (0 = default)
sc = {"cfg2": scenes.config2, "cfg4": scenes.config4, "cfg1": scenes.config1}[cfg](1920, 1080, spp_half)
flat = sc.flatten()
A, B = Context(0), Context(0)
for c in (A, B):
    c.upload(flat)
    c.reserve(sc.camera, 1, msb)
    c.render(sc.camera, seed=1, want_f32=False, want_u8=False, max_state_bytes=msb)
def timed(fn, reps=3):
    best = 1e9
    for _ in range(reps):
        t0 = time.perf_counter(); fn(); best = min(best, (time.perf_counter() - t0) * 1e3)
    return best
seq = timed(lambda: (A.render(sc.camera, seed=1, want_f32=False, want_u8=False, max_state_bytes=msb), B.render(sc.camera, seed=2, want_f32=False, want_u8=False, max_state_bytes=msb)))
one = timed(lambda: A.render(sc.camera, seed=1, want_f32=False, want_u8=False, max_state_bytes=msb))
print(f"RES {cfg} half = {spp_half} spp, budget {msb >> 30} GiB per context: one half {one:.2f} ms, two in a row {seq:.2f} ms", flush=True)
for delay in (0.0, 10.0, 20.0, 25.0, 28.0, 31.0, 34.0):
    def pair():
        tb = threading.Thread(target=lambda: (time.sleep(delay * 1e-3), B.render(sc.camera, seed=2, want_f32=False, want_u8=False, max_state_bytes=msb)))
        tb.start(); A.render(sc.camera, seed=1, want_f32=False, want_u8=False, max_state_bytes=msb); tb.join()
    print(f"RES {cfg} B starts {delay:.0f} ms after A: pair {timed(pair):.2f} ms", flush=True)
A.close(); B.close()
