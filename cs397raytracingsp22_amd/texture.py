"""Host-side mirror of src/util/texture.rs: `Texture` holds a decoded RGB8 image.
`sample` (texture.rs:26-32) runs on the GPU; decoding (texture.rs:16-25) is load-time."""
from __future__ import annotations

import numpy as np


class Texture:
    def __init__(self, rgb: np.ndarray):
        rgb = np.ascontiguousarray(rgb, dtype=np.uint8)
        assert rgb.ndim == 3 and rgb.shape[2] == 3
        self.img = rgb                      # [H, W, 3], row 0 = top of the image

    @property
    def width(self):
        return self.img.shape[1]

    @property
    def height(self):
        return self.img.shape[0]

    @staticmethod
    def load_from_file(file_name: str):
        """Texture::load_from_file (texture.rs:16-25): None when the file cannot be opened
        or decoded (the reference swallows the error the same way).  `.npy` holds a
        pre-decoded [H,W,3] uint8 array; other formats go through PIL when present."""
        try:
            if file_name.endswith(".npy"):
                return Texture(np.load(file_name, allow_pickle=False))
            from PIL import Image
            with Image.open(file_name) as im:
                return Texture(np.asarray(im.convert("RGB")))
        except Exception:
            return None
