"""C++ host texture decode (cs397raytracingsp22_amd/host/texture.hpp, the mirror of Texture::load_from_file,
texture.rs:16-25): PNG, TGA and JPEG (baseline + progressive) decode byte for byte like PIL's `.convert("RGB")` — for PNG /
TGA that is what the reference's `get_pixel(..).to_rgb()` yields; for JPEG it is libjpeg's result, from which the `image`
crate's own decoder may differ by 1-2 LSB.  Anything it cannot decode gives nullopt, the reference's None."""
import os
import subprocess

import numpy as np
import pytest
from PIL import Image

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_TEX = "/root/reference/texture"


@pytest.fixture(scope="module")
def decoder(tmp_path_factory):
    exe = tmp_path_factory.mktemp("tex") / "texture_decode_check"
    subprocess.run(["g++", "-O2", "-std=c++17", "-Wall", os.path.join(ROOT, "tests", "cpp", "texture_decode_check.cpp"),
                    "-o", str(exe), "-lz"], check=True)
    return str(exe)


def decode(decoder, path, tmp_path):
    out = tmp_path / "out.raw"
    r = subprocess.run([decoder, str(path), str(out)])
    if r.returncode == 3:
        return None
    assert r.returncode == 0
    raw = open(out, "rb").read()
    head, body = raw.split(b"\n", 1)
    w, h = (int(v) for v in head.split())
    return np.frombuffer(body, np.uint8).reshape(h, w, 3)


def pil_rgb(path):
    return np.asarray(Image.open(path).convert("RGB"))


@pytest.mark.parametrize("name", ["green.png", "white.png", "normal_test.png"])
def test_reference_png_assets_decode_exactly(decoder, tmp_path, name):
    path = os.path.join(REF_TEX, name)
    if not os.path.exists(path):
        pytest.skip("reference assets are not on this machine")
    assert np.array_equal(decode(decoder, path, tmp_path), pil_rgb(path))


def synthetic(seed, w, h, mode):
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:h, 0:w]
    base = np.stack([(xx * 7 + yy * 3) % 256, (xx * 2 + yy * 11) % 256, (xx + yy) % 256, (xx * 5) % 256], axis=2).astype(np.uint8)
    base[::3, ::4] = rng.integers(0, 256, base[::3, ::4].shape, dtype=np.uint8)
    if mode == "RGB":
        return Image.fromarray(base[:, :, :3], "RGB")
    if mode == "RGBA":
        return Image.fromarray(base, "RGBA")
    if mode == "L":
        return Image.fromarray(base[:, :, 0], "L")
    if mode == "LA":
        return Image.fromarray(base[:, :, :2], "LA")
    if mode == "P":
        return Image.fromarray(base[:, :, :3], "RGB").quantize(colors=200)
    if mode == "P4":
        return Image.fromarray(base[:, :, :3], "RGB").quantize(colors=16)
    if mode == "1":
        return Image.fromarray(base[:, :, 0] > 127)
    raise ValueError(mode)


@pytest.mark.parametrize("mode", ["RGB", "RGBA", "L", "LA", "P", "P4", "1"])
@pytest.mark.parametrize("size", [(64, 48), (37, 53), (1, 1), (9, 3)])
def test_png_colour_types_and_ragged_sizes(decoder, tmp_path, mode, size):
    img = synthetic(hash(mode) % 1000, size[0], size[1], mode)
    p = tmp_path / f"t_{mode}.png"
    kw = {"bits": 4} if mode == "P4" else {}
    img.save(p, optimize=(size[0] > 9), **kw)
    assert np.array_equal(decode(decoder, p, tmp_path), pil_rgb(p))


def test_png_interlaced(decoder, tmp_path):
    """Adam7: PIL cannot write it, so one is assembled here by hand (7 passes, filter 0) from a known image."""
    import struct
    import zlib
    w, h = 23, 17
    img = np.asarray(synthetic(5, w, h, "RGB"))
    passes = [(0, 0, 8, 8), (4, 0, 8, 8), (0, 4, 4, 8), (2, 0, 4, 4), (0, 2, 2, 4), (1, 0, 2, 2), (0, 1, 1, 2)]
    raw = b""
    for x0, y0, dx, dy in passes:
        sub = img[y0::dy, x0::dx]
        if sub.size:
            for row in sub:
                raw += b"\x00" + row.tobytes()

    def chunk(t, d):
        return struct.pack(">I", len(d)) + t + d + struct.pack(">I", zlib.crc32(t + d) & 0xffffffff)
    png = b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, 2, 0, 0, 1)) + chunk(b"IDAT", zlib.compress(raw)) + chunk(b"IEND", b"")
    p = tmp_path / "a7.png"
    p.write_bytes(png)
    assert np.array_equal(pil_rgb(p), img)                    # PIL reads it: the hand-made file is a valid PNG
    assert np.array_equal(decode(decoder, p, tmp_path), img)
    bad = bytearray(png)
    bad[60] ^= 0x55                                            # corrupt the IDAT payload: CRC mismatch -> None
    q = tmp_path / "bad.png"
    q.write_bytes(bytes(bad))
    assert decode(decoder, q, tmp_path) is None


@pytest.mark.parametrize("mode,rle", [("RGB", False), ("RGB", True), ("RGBA", False), ("RGBA", True), ("L", False), ("L", True), ("P", False)])
def test_tga_variants(decoder, tmp_path, mode, rle):
    """The drone's five maps are TGA files (absent from the reference, .MISSING_LARGE_BLOBS): true-colour / grey / colour-mapped,
    raw and run-length encoded, written by PIL (bottom-left origin) and by hand (top-left origin)."""
    img = synthetic(11, 61, 45, mode)
    p = tmp_path / f"t_{mode}_{rle}.tga"
    img.save(p, compression="tga_rle" if rle else None)
    assert np.array_equal(decode(decoder, p, tmp_path), pil_rgb(p))
    if mode == "RGB" and not rle:                              # same pixels, top-left origin flag set and rows reordered
        d = bytearray(p.read_bytes())
        w, h = 61, 45
        body = np.frombuffer(bytes(d[18:18 + w * h * 3]), np.uint8).reshape(h, w, 3)[::-1]
        d[17] |= 0x20
        d[18:18 + w * h * 3] = body.tobytes()
        q = tmp_path / "top.tga"
        q.write_bytes(bytes(d))
        assert np.array_equal(decode(decoder, q, tmp_path), pil_rgb(p))


def test_undecodable_files_give_none(decoder, tmp_path):
    p = tmp_path / "noise.bin"
    p.write_bytes(bytes(range(7)))
    assert decode(decoder, p, tmp_path) is None
    assert decode(decoder, tmp_path / "does_not_exist.png", tmp_path) is None
    cmyk = tmp_path / "cmyk.jpg"
    synthetic(1, 16, 16, "RGB").convert("CMYK").save(cmyk)
    assert decode(decoder, cmyk, tmp_path) is None             # four components: documented as not decoded
    trunc = tmp_path / "trunc.jpg"
    good = tmp_path / "good.jpg"
    synthetic(1, 64, 64, "RGB").save(good)
    trunc.write_bytes(good.read_bytes()[:200])
    assert decode(decoder, trunc, tmp_path) is None


@pytest.mark.parametrize("name", ["earthmap.jpg", "normal_test.jpg", "magenta.jpg"])
def test_reference_jpeg_assets_decode_like_libjpeg(decoder, tmp_path, name):
    """Baseline 4:4:4, baseline 4:2:0 and PROGRESSIVE 4:2:0: bit-identical to PIL (libjpeg-turbo: islow IDCT, fancy upsampling)."""
    path = os.path.join(REF_TEX, name)
    if not os.path.exists(path):
        pytest.skip("reference assets are not on this machine")
    assert np.array_equal(decode(decoder, path, tmp_path), pil_rgb(path))


@pytest.mark.parametrize("size", [(64, 48), (37, 53), (1, 1), (17, 9), (130, 66)])
@pytest.mark.parametrize("kw", [dict(subsampling=0), dict(subsampling=1), dict(subsampling=2), dict(subsampling=2, progressive=True),
                                dict(subsampling=0, progressive=True, quality=95), dict(subsampling=2, quality=30, optimize=True),
                                dict(grey=True), dict(grey=True, progressive=True), dict(subsampling=2, restart_marker_blocks=3)])
def test_jpeg_variants(decoder, tmp_path, size, kw):
    kw = dict(kw)
    img = synthetic(7, size[0], size[1], "L" if kw.pop("grey", False) else "RGB")
    p = tmp_path / "t.jpg"
    try:
        img.save(p, **kw)
    except (TypeError, OSError):
        pytest.skip("this Pillow cannot write that variant")
    got = decode(decoder, p, tmp_path)
    assert got is not None
    assert np.array_equal(got, pil_rgb(p))


@pytest.fixture(scope="module")
def decoder_asan(tmp_path_factory):
    """The same decoder built with AddressSanitizer + UBSan (CPU build: sanitizers never run on the GPU box's device code)."""
    libasan = subprocess.run(["g++", "-print-file-name=libasan.so"], capture_output=True, text=True).stdout.strip()
    if not os.path.isabs(libasan) or not os.path.exists(libasan):
        pytest.skip("libasan not available")
    exe = tmp_path_factory.mktemp("texasan") / "texture_decode_check_asan"
    subprocess.run(["g++", "-O1", "-g", "-std=c++17", "-fsanitize=address,undefined", "-fno-sanitize-recover=all",
                    os.path.join(ROOT, "tests", "cpp", "texture_decode_check.cpp"), "-o", str(exe), "-lz"], check=True)
    return str(exe)


def jpeg_with(sof_wh=None, cut_after_sos_len=False):
    """A small valid JPEG, then its header tampered with: SOF0 width/height overwritten, or the file cut right behind an SOS
    segment length of 2 (the component count would be read one byte past the end)."""
    import io
    buf = io.BytesIO()
    synthetic(3, 32, 32, "RGB").save(buf, format="JPEG")
    d = bytearray(buf.getvalue())
    if sof_wh is not None:
        k = d.index(b"\xff\xc0")
        d[k + 5:k + 7] = sof_wh[1].to_bytes(2, "big")
        d[k + 7:k + 9] = sof_wh[0].to_bytes(2, "big")
    if cut_after_sos_len:
        k = d.index(b"\xff\xda")
        d = d[:k + 2] + b"\x00\x02"
    return bytes(d)


def test_hostile_headers_are_refused_without_allocating_or_overreading(decoder_asan, tmp_path):
    cases = {
        "sos_cut.jpg": jpeg_with(cut_after_sos_len=True),                 # SOS of length 2 ending at EOF
        "huge_sof.jpg": jpeg_with(sof_wh=(65535, 65535)),                 # 4.3 G pixels claimed by a 1 KB file
        "wide_sof.jpg": jpeg_with(sof_wh=(16000, 8000)),                  # under the pixel cap, far beyond what the bytes can hold
    }
    # PNG: a 32768 x 32768 header over a few bytes of IDAT (CRC-correct chunks)
    import struct
    import zlib

    def chunk(t, b):
        return struct.pack(">I", len(b)) + t + b + struct.pack(">I", zlib.crc32(t + b) & 0xffffffff)
    cases["huge.png"] = (b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", 32768, 32768, 8, 2, 0, 0, 0)) +
                         chunk(b"IDAT", zlib.compress(b"\x00" * 64)) + chunk(b"IEND", b""))
    # TGA: 65535 x 65535 true-colour, raw and RLE, 18-byte header + a few bytes
    for name, itype in (("huge_raw.tga", 2), ("huge_rle.tga", 10)):
        h = bytearray(18)
        h[2] = itype
        h[12:14] = (65535).to_bytes(2, "little")
        h[14:16] = (65535).to_bytes(2, "little")
        h[16] = 24
        cases[name] = bytes(h) + b"\x7f" + b"\x01\x02\x03" * 4
    for name, data in cases.items():
        p = tmp_path / name
        p.write_bytes(data)
        out = tmp_path / "o.raw"
        r = subprocess.run([decoder_asan, str(p), str(out)], capture_output=True, text=True,
                           env=dict(os.environ, ASAN_OPTIONS="detect_leaks=0:abort_on_error=1:max_allocation_size_mb=512"))
        assert r.returncode == 3, (name, r.returncode, r.stderr[-800:])   # 3 = nullopt (None); anything else is a crash or a decode
