"""Progressive / resumable rendering on one GPU (output stage).

The reference sums the `aa_sample_count` samples of a pixel in one loop (tracing.rs:233-241).
`ProgressiveRender` runs that loop in slices — samples [0, a), [a, b), ... — keeping the running
per-pixel sums in a device buffer, so that
  * an image can be shown after every slice (`preview()`: sums / samples so far), and
  * the state can be written to disk and picked up later, also by another process
    (`save()` / `ProgressiveRender.resume()`).
Because every sample owns its RNG stream and the slices are added in sample order, the final image
is bit-identical to a one-call render.  torch supplies the device buffers; compute is the HIP library.
"""
from __future__ import annotations

import hashlib

import numpy as np

from . import abi
from .dist import TILE_PIXELS, compact_index, tiles_padded


def _camera_key(cam, seed: int) -> str:
    pod = cam.to_pod()
    return hashlib.sha256(bytes(pod) + int(seed).to_bytes(4, "little")).hexdigest()


class ProgressiveRender:
    def __init__(self, ctx, camera, seed: int = 1, want_sig: bool = False):
        import torch
        if not torch.cuda.is_available():
            raise RuntimeError("torch cannot see the GPU: import torch BEFORE creating a Context")
        self.torch, self.ctx, self.cam, self.seed = torch, ctx, camera, seed
        dev = torch.device(f"cuda:{ctx.device}")
        W, H = camera.screen_width, camera.screen_height
        self.padded = tiles_padded(W, H, 1)
        n = self.padded * TILE_PIXELS
        self.accum = torch.zeros((n, 4), dtype=torch.float32, device=dev)
        self.compact = torch.zeros((n, 3), dtype=torch.float32, device=dev)
        self.sig = torch.zeros((n,), dtype=torch.int32, device=dev) if want_sig else None
        self.done = 0                                   # samples per pixel accumulated so far
        _, idx = compact_index(W, H, 1)
        self._idx = torch.from_numpy(idx.astype(np.int64)).to(dev)
        ctx.reserve(camera, 1)

    @property
    def finished(self) -> bool:
        return self.done == self.cam.aa_sample_count

    def advance(self, samples: int):
        """Trace the next `samples` samples of every pixel (clamped to what is left)."""
        end = min(self.cam.aa_sample_count, self.done + int(samples))
        if end == self.done:
            return None
        stream = self.torch.cuda.current_stream(self.accum.device).cuda_stream
        st = self.ctx.render_samples_device(self.cam, self.done, end, self.accum.data_ptr(), self.compact.data_ptr(),
                                            self.sig.data_ptr() if self.sig is not None else None, seed=self.seed,
                                            stream=stream)
        self.done = end
        return st

    def preview(self) -> np.ndarray:
        """[H, W, 3] f32 mean of the samples accumulated so far (tracing.rs:241 with n = done)."""
        if self.done == 0:
            raise RuntimeError("nothing accumulated yet")
        img = self.accum[:, :3][self._idx] / float(self.done)
        return img.cpu().numpy()

    def result(self):
        """Final ([H,W,3] f32 means, [H,W] u32 signatures or None): what a one-call render returns."""
        if not self.finished:
            raise RuntimeError(f"{self.done} of {self.cam.aa_sample_count} samples accumulated")
        img = self.compact[self._idx].cpu().numpy()
        sig = self.sig[self._idx].cpu().numpy().view(np.uint32) if self.sig is not None else None
        return img, sig

    # ---- checkpoint / resume ----
    def save(self, path: str):
        self.torch.cuda.synchronize(self.accum.device)
        np.savez(path, accum=self.accum.cpu().numpy(), done=np.uint32(self.done), seed=np.uint32(self.seed),
                 key=np.array(_camera_key(self.cam, self.seed)))

    @classmethod
    def resume(cls, ctx, camera, path: str, want_sig: bool = False):
        z = np.load(path, allow_pickle=False)
        seed = int(z["seed"])
        if str(z["key"]) != _camera_key(camera, seed):
            raise ValueError("checkpoint was made with a different camera or seed")
        pr = cls(ctx, camera, seed=seed, want_sig=want_sig)
        acc = z["accum"]
        if acc.shape != tuple(pr.accum.shape):
            raise ValueError("checkpoint accumulator has the wrong shape")
        pr.accum.copy_(pr.torch.from_numpy(acc))
        pr.done = int(z["done"])
        return pr
