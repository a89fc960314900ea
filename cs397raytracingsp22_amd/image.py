"""Output stage of the reference's run(): `RgbImage::save_with_format("render.png", ImageFormat::Png)`
(tracing.rs:546).  Load/store work outside the accelerated path; a dependency-free PNG encoder
(zlib + CRC from the standard library) so the host mirror can finish what run() does."""
from __future__ import annotations

import struct
import zlib

import numpy as np


def save_png(path: str, rgb: np.ndarray) -> None:
    """Write an [H, W, 3] uint8 array as an 8-bit truecolour PNG (filter 0 on every scanline)."""
    rgb = np.ascontiguousarray(rgb, dtype=np.uint8)
    assert rgb.ndim == 3 and rgb.shape[2] == 3
    h, w, _ = rgb.shape
    raw = np.empty((h, 1 + 3 * w), np.uint8)
    raw[:, 0] = 0
    raw[:, 1:] = rgb.reshape(h, 3 * w)

    def chunk(tag: bytes, data: bytes) -> bytes:
        return struct.pack(">I", len(data)) + tag + data + struct.pack(">I", zlib.crc32(tag + data) & 0xFFFFFFFF)

    with open(path, "wb") as fh:
        fh.write(b"\x89PNG\r\n\x1a\n")
        fh.write(chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, 2, 0, 0, 0)))
        fh.write(chunk(b"IDAT", zlib.compress(raw.tobytes(), 6)))
        fh.write(chunk(b"IEND", b""))
