#!/usr/bin/env python3
"""A long run of the random-scene parity tests (tests/test_gpu_fuzz.py) over seeds the test suite does not use:

    python tools/fuzz_campaign.py [first_seed] [count]

Runs test_random_scene_parity and test_random_scene_random_modes for every seed in [first_seed, first_seed + count)
against the oracle on the GPU box, prints one line per failure and a summary.  A one-off confidence run, not part of the suite.
"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402,F401
from cs397raytracingsp22_amd import Context  # noqa: E402
from oracle import orc_py  # noqa: E402
import test_gpu_fuzz as F  # noqa: E402


def self_consistency(ctx, seed):
    """The optimisations against each other on one random scene: forced two-stage / reference walk / no tile masks, a tiny
    state budget (many sample batches) and a virtual 3-rank tiling must all give the default render's signatures and image."""
    import numpy as np
    from cs397raytracingsp22_amd import abi
    from test_gpu_tiles import assemble
    sc = F.random_scene(50000 + seed)
    cam = sc.camera
    ctx.upload(sc.flatten())
    f0, _, s0, _ = ctx.render(cam, seed=seed, want_u8=False, want_sig=True)
    from cs397raytracingsp22_amd import dist as pdist
    # the smallest budget the pipeline accepts: two samples per padded pixel (300 B per path covers the two-stage buffers too)
    tiny = 2 * pdist.tiles_padded(cam.screen_width, cam.screen_height, 1) * pdist.TILE_PIXELS * 300
    for name, kw in (("two-stage", dict(flags=abi.MI_OPT_TWO_STAGE)), ("reference-walk", dict(flags=abi.MI_OPT_REFERENCE_WALK)),
                     ("no-tile-masks", dict(flags=abi.MI_OPT_NO_TILE_MASKS)), ("small-batches", dict(max_state_bytes=tiny)),
                     ("voted", dict(variant=abi.MI_VARIANT_VOTED))):
        f1, _, s1, _ = ctx.render(cam, seed=seed, want_u8=False, want_sig=True, **kw)
        assert np.array_equal(s0, s1), f"{name}: {int((s0 != s1).sum())} signatures differ"
        assert np.array_equal(f0, f1, equal_nan=True), f"{name}: image differs"
    if cam.lens_radius == 0.0:
        img, _, _ = assemble(ctx, cam, 3, seed=seed)
        assert np.array_equal(img, f0, equal_nan=True), "3 virtual ranks: image differs"


def main():
    first = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    count = int(sys.argv[2]) if len(sys.argv) > 2 else 300
    orc_py.build()
    orc_py.load()
    ctx = Context(0)
    bad = 0
    t0 = time.time()
    for seed in range(first, first + count):
        for fn in (F.test_random_scene_parity, F.test_random_scene_random_modes, lambda c, o, sd: self_consistency(c, sd)):
            try:
                fn(ctx, orc_py, seed)
            except AssertionError as e:
                bad += 1
                print(f"FAIL {getattr(fn, '__name__', 'self_consistency')} seed {seed}: {str(e)[:200]}", flush=True)
        if (seed - first) % 50 == 49:
            print(f"... {seed - first + 1} seeds, {bad} failures, {time.time() - t0:.0f} s", flush=True)
    print(f"fuzz campaign: seeds [{first}, {first + count}), 3 tests each: {bad} failures in {time.time() - t0:.0f} s", flush=True)
    ctx.close()
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
