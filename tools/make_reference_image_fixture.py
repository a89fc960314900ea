#!/usr/bin/env python3
"""Reduce the reference's published render of its HEAD scene (render.png, 800x800 — an
OUTPUT of the real Rust program, the only one whose scene code is in HEAD) to a 50x50
grid of block means and store it as tests/golden/reference_render_50x50.npy.
Data, not source.  Run in the build container:  python tools/make_reference_image_fixture.py"""
import os
import sys

import numpy as np
from PIL import Image

REF = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
img = np.asarray(Image.open(os.path.join(REF, "render.png")).convert("RGB")).astype(np.float32) / 255.0
n = 50
h = img.shape[0] // n
small = img[:h * n, :h * n].reshape(n, h, n, h, 3).mean(axis=(1, 3)).astype(np.float32)
np.save(os.path.join(OUT, "reference_render_50x50.npy"), small)
print(small.shape, small.mean(axis=(0, 1)))
