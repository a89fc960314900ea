#!/usr/bin/env python3
"""Measure the per-sample work counters of a configuration with the CPU oracle
(box tests, triangle tests, segments ... per camera sample) and commit them as
tests/golden/workcounts_<cfg>.json.  bench.py turns them into the ALGORITHMIC bytes
per sample that `roofline.achieved` is computed from (DESIGN.md "Roofline").

Sample: `bands` full-width bands of `rows` rows, evenly spaced over the image, all spp.
"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cs397raytracingsp22_amd import scenes  # noqa: E402
from oracle import orc_py  # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")


def bands_of(height, bands, rows):
    step = height / bands
    return [min(height - rows, int(i * step + step / 2 - rows / 2)) for i in range(bands)]


def measure(name, sc, bands=8, rows=4):
    flat = sc.flatten()
    o = orc_py.OracleScene(flat)
    cam = sc.camera
    tot = {}
    for y0 in bands_of(cam.screen_height, bands, rows):
        _, _, _, c = o.render(cam, seed=1, window=(0, y0, cam.screen_width, rows), want_u8=False, want_sig=False,
                              want_counters=True)
        for k, v in c.items():
            tot[k] = tot.get(k, 0) + v
    per = {k: v / tot["samples"] for k, v in tot.items() if k != "samples"}
    # bytes of the linear object list one segment dereferences (SURVEY.md §8d B_list)
    d = flat.desc
    kinds = [d.objects[i].kind for i in range(d.n_objects)]
    b_list = sum({0: 16, 1: 36, 2: 24, 3: 20, 4: 128}[k] for k in kinds)
    rec = {"config": name, "width": cam.screen_width, "height": cam.screen_height, "spp": cam.aa_sample_count,
           "path_depth": cam.path_depth, "sample": f"{bands} bands x {rows} rows, full width, all spp, seed 1",
           "samples_measured": tot["samples"], "per_sample": per, "b_list_bytes": b_list}
    with open(os.path.join(OUT, f"workcounts_{name}.json"), "w") as fh:
        json.dump(rec, fh, indent=1, sort_keys=True)
    print(name, json.dumps(per))


if __name__ == "__main__":
    which = sys.argv[1:] or ["cfg1", "cfg2"]
    if "cfg1" in which:
        measure("cfg1", scenes.config1(), bands=8, rows=8)
    if "cfg2" in which:
        measure("cfg2", scenes.config2(), bands=8, rows=2)
    # the counters are per camera sample: a lower spp than the configuration's measures the same ratios
    if "cfg4" in which:
        measure("cfg4", scenes.config4(spp=16), bands=8, rows=2)
    if "cfg5" in which:
        measure("cfg5", scenes.config5(spp=64), bands=8, rows=2)
    if "head" in which:
        measure("head", scenes.head_scene(800, 800, 16, textures=scenes.load_asset_textures()), bands=8, rows=2)
