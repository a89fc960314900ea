"""The C-ABI library loads and exports every symbol include/mi_rt.h declares; the ctypes
mirror has the C layout; the product fails loudly (no CPU fallback) without a GPU.
No compute calls here: CPU-only."""
import ctypes as C
import os
import re
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "mi_rt.h")


def declared_functions():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(mi_[a-z_0-9]+)\s*\(", src)))


def test_header_declares_expected_entry_points():
    from cs397raytracingsp22_amd import abi
    assert declared_functions() == sorted(abi.EXPORTS)


def test_library_exports_every_declared_symbol():
    from cs397raytracingsp22_amd import abi
    lib = abi.load()
    for name in declared_functions():
        assert hasattr(lib, name), name
    assert lib.mi_abi_version() == abi.MI_RT_ABI_VERSION == 5


def test_ctypes_layout_matches_c(tmp_path):
    from cs397raytracingsp22_amd import abi
    names = ["mi_material", "mi_object", "mi_sphere", "mi_triangle", "mi_plane", "mi_volume", "mi_texture", "mi_mesh",
             "mi_scene_desc", "mi_camera_desc", "mi_render_opts", "mi_stats"]
    prog = '#include <stdio.h>\n#include "mi_rt.h"\nint main(void){' + "".join(
        f'printf("{n} %zu\\n", sizeof({n}));' for n in names) + "return 0;}\n"
    src = tmp_path / "sz.c"
    src.write_text(prog)
    exe = tmp_path / "sz"
    subprocess.run(["gcc", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)], check=True)
    out = subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout
    sizes = dict(line.split() for line in out.strip().splitlines())
    for n in names:
        assert C.sizeof(getattr(abi, n)) == int(sizes[n]), n


def test_no_cpu_fallback_without_gpu():
    """On a box without a GPU the product must fail loudly, never render on the CPU."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    from cs397raytracingsp22_amd import Context, abi
    with pytest.raises(abi.MiError) as ei:
        Context(0)
    assert ei.value.code == abi.MI_ERR_NO_DEVICE
    assert "no CPU fallback" in str(ei.value)


def test_product_never_imports_oracle():
    """oracle/ is test infrastructure: nothing in the product package may reference it."""
    pkg = os.path.join(ROOT, "cs397raytracingsp22_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".h", ".hpp", ".sh")):
                text = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "orc_" not in text and "liborc" not in text and "from oracle" not in text and "import oracle" not in text, f


def test_abi_version_is_one_number_everywhere():
    """include/mi_rt.h, the Python mirror, the library and __graft_entry__.build() agree."""
    import re
    from cs397raytracingsp22_amd import abi
    hdr = open(os.path.join(ROOT, "include", "mi_rt.h")).read()
    assert int(re.search(r"#define\s+MI_RT_ABI_VERSION\s+(\d+)", hdr).group(1)) == abi.MI_RT_ABI_VERSION
    assert abi.load().mi_abi_version() == abi.MI_RT_ABI_VERSION
    entry = open(os.path.join(ROOT, "__graft_entry__.py")).read()
    assert "abi.MI_RT_ABI_VERSION" in entry            # build() checks against the mirror, not a literal


def test_rust_shim_declares_every_entry_point():
    """rust/src/mi_rt.rs cannot be compiled here (no cargo / rustc): at least keep it complete — every function of the
    header is declared, and the PODs carry the header's field counts."""
    src = open(os.path.join(ROOT, "rust", "src", "util", "mi_rt.rs")).read()
    declared = set(re.findall(r"pub fn (mi_[a-z_0-9]+)\(", src))
    assert declared == set(declared_functions())
    assert "pub flags: u32" in src and "pub max_state_bytes: u64" in src          # mi_render_opts, ABI 3
    assert "pub boundary_kind: i32" in src and "pub boundary_objects: *const mi_object" in src      # ABI 4
