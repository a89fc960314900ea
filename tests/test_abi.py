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


RUST_DIR = os.path.join(ROOT, "rust", "src", "util")


def _strip_c_comments(src):
    return re.sub(r"/\*.*?\*/", "", src, flags=re.S)


def _strip_rust(src):
    """Rust source without comments, string literals and char literals (enough for the structural checks below)."""
    out, i, n = [], 0, len(src)
    while i < n:
        c = src[i]
        if src.startswith("//", i):
            while i < n and src[i] != "\n":
                i += 1
        elif src.startswith("/*", i):
            i = src.index("*/", i) + 2
        elif c == '"':
            i += 1
            while src[i] != '"':
                i += 2 if src[i] == "\\" else 1
            i += 1
            out.append('""')
        elif c == "'" and re.match(r"'(\\.|[^\\'])'", src[i:]):
            i += re.match(r"'(\\.|[^\\'])'", src[i:]).end()
            out.append("' '")
        else:
            out.append(c)
            i += 1
    return "".join(out)


C_SCALAR = {"int32_t": "i32", "uint32_t": "u32", "uint64_t": "u64", "float": "f32", "uint8_t": "u8", "int": "c_int",
            "void": "c_void", "char": "c_char"}


def _c_type_to_rust(ctype, array):
    """`const float*` -> `*const f32`, `float` + [3] -> `[f32; 3]`, `mi_ctx**` -> `*mut *mut mi_ctx`."""
    t = ctype.strip()
    const = t.startswith("const ")
    if const:
        t = t[len("const "):].strip()
    stars = t.count("*")
    base = t.replace("*", "").strip()
    r = C_SCALAR.get(base, base)
    for k in range(stars):
        r = ("*const " if (const and k == 0) else "*mut ") + r
    if array is not None:
        r = f"[{r}; {array}]"
    return r


def _header_structs():
    src = _strip_c_comments(open(HEADER).read())
    structs = {}
    for body, name in re.findall(r"typedef\s+struct\s+\w+\s*\{(.*?)\}\s*(\w+)\s*;", src, flags=re.S):
        fields = []
        for decl in body.split(";"):
            decl = decl.strip()
            if not decl:
                continue
            m = re.match(r"(.+?)\s*(\w+)\s*(?:\[(\d+)\])?$", decl, flags=re.S)
            ctype, fname, arr = m.group(1), m.group(2), m.group(3)
            fields.append((fname, _c_type_to_rust(" ".join(ctype.split()), arr)))
        structs[name] = fields
    return structs


def _rust_structs():
    src = _strip_rust(open(os.path.join(RUST_DIR, "mi_rt.rs")).read())
    structs = {}
    for m in re.finditer(r"#\[repr\(C\)\]\s*(?:#\[derive\([^\]]*\)\]\s*)?pub\s+struct\s+(\w+)\s*\{(.*?)\}", src, flags=re.S):
        body = m.group(2)
        fields, depth, cur = [], 0, ""
        for ch in body:                                   # split at top-level commas ([f32; 3] holds none, but stay safe)
            if ch in "[(<":
                depth += 1
            elif ch in "])>":
                depth -= 1
            if ch == "," and depth == 0:
                fields.append(cur)
                cur = ""
            else:
                cur += ch
        fields.append(cur)
        parsed = []
        for f in fields:
            f = f.strip()
            if not f:
                continue
            fm = re.match(r"pub\s+(\w+)\s*:\s*(.+)$", f, flags=re.S)
            assert fm, (m.group(1), f)
            parsed.append((fm.group(1), " ".join(fm.group(2).split())))
        structs[m.group(1)] = parsed
    return structs


def test_rust_structs_match_the_header_field_by_field():
    """Every #[repr(C)] struct of rust/src/util/mi_rt.rs against the typedef of the same name in include/mi_rt.h: same fields,
    same order, same types (int32_t = i32, float[3] = [f32; 3], const T* = *const T, ...).  With #[repr(C)] that is the C layout."""
    c, r = _header_structs(), _rust_structs()
    expect = {"mi_material", "mi_object", "mi_sphere", "mi_triangle", "mi_plane", "mi_volume", "mi_texture", "mi_mesh",
              "mi_scene_desc", "mi_camera_desc", "mi_render_opts", "mi_stats"}
    assert expect <= set(c) and set(r) == expect, (sorted(c), sorted(r))
    for name in sorted(expect):
        assert r[name] == c[name], f"{name}:\n rust {r[name]}\n    C {c[name]}"


def _header_prototypes():
    src = _strip_c_comments(open(HEADER).read())
    src = src[src.index("typedef struct mi_ctx mi_ctx;"):]
    protos = {}
    for ret, name, args in re.findall(r"([\w\s\*]+?)\b(mi_[a-z_0-9]+)\s*\(([^)]*)\)\s*;", src):
        ret = " ".join(ret.split())
        a = []
        for arg in args.split(","):
            arg = " ".join(arg.split())
            if arg in ("void", ""):
                continue
            m = re.match(r"(.+?)\s*(\w+)$", arg)
            a.append(_c_type_to_rust(m.group(1), None))
        protos[name] = (None if ret == "void" else _c_type_to_rust(ret, None), a)
    return protos


def _rust_prototypes():
    src = _strip_rust(open(os.path.join(RUST_DIR, "mi_rt.rs")).read())
    block = re.search(r'extern\s+""\s*\{(.*?)\n\}', src, flags=re.S).group(1)
    protos = {}
    for name, args, ret in re.findall(r"pub\s+fn\s+(mi_\w+)\s*\(([^)]*)\)\s*(?:->\s*([^;]+))?;", block):
        a = [" ".join(x.split(":", 1)[1].split()) for x in args.split(",") if x.strip()]
        protos[name] = (" ".join(ret.split()) if ret else None, a)
    return protos


def test_rust_extern_block_matches_the_header_prototypes():
    """Every function of the header is declared in the extern "C" block with the same argument count and types and the same
    return type (int = c_int; int32_t = i32; T* = *mut T; const T* = *const T)."""
    c, r = _header_prototypes(), _rust_prototypes()
    assert set(c) == set(r) == set(declared_functions())

    def norm(t):          # the header writes `int` and `int32_t` for what Rust may call c_int / i32: one 32-bit int on this ABI
        return None if t is None else t.replace("c_int", "i32")
    for name in sorted(c):
        assert norm(r[name][0]) == norm(c[name][0]), (name, r[name][0], c[name][0])
        assert [norm(t) for t in r[name][1]] == [norm(t) for t in c[name][1]], (name, r[name][1], c[name][1])


def test_rust_constants_match_the_header():
    hdr = open(HEADER).read()
    src = open(os.path.join(RUST_DIR, "mi_rt.rs")).read()
    rust = {k: int(v) for k, v in re.findall(r"pub const (MI_\w+): \w+ = (\d+);", src)}
    cvals = {k: int(v.rstrip("u")) for k, v in re.findall(r"#define\s+(MI_\w+)\s+(\d+u?)\b", hdr)}
    cvals.update({k: int(v) for k, v in re.findall(r"\b(MI_(?:MAT|OBJ)_\w+)\s*=\s*(\d+)", hdr)})
    assert rust["MI_RT_ABI_VERSION"] == cvals["MI_RT_ABI_VERSION"]
    for k, v in rust.items():
        assert k in cvals and cvals[k] == v, k
    for k in cvals:
        if k.startswith(("MI_MAT_", "MI_OBJ_", "MI_PROJ_", "MI_SHADE_", "MI_OPT_")):
            assert k in rust, k


def test_rust_patch_set_is_complete_and_well_formed():
    """The ten flatten implementations + Camera + Texture + Scene::render_to_image exist as source, every file's brackets
    balance, and every mi_* struct literal names exactly the fields of that struct (a literal that forgets or misnames a field
    would not compile)."""
    files = {f: open(os.path.join(RUST_DIR, f)).read() for f in sorted(os.listdir(RUST_DIR)) if f.endswith(".rs")}
    assert set(files) == {"mi_rt.rs", "geometry_flatten.rs", "materials_flatten.rs", "texture_flatten.rs", "tracing_flatten.rs"}
    for name, src in files.items():
        code = _strip_rust(src)
        stack = []
        for ch in code:
            if ch in "([{":
                stack.append(ch)
            elif ch in ")]}":
                assert stack and "([{".index(stack.pop()) == ")]}".index(ch), name
        assert not stack, name
    geo, mat, tra, tex = (files[k] for k in ("geometry_flatten.rs", "materials_flatten.rs", "tracing_flatten.rs", "texture_flatten.rs"))
    for t in ("Sphere", "Triangle", "Plane", "ConvexVolume", "StaticMesh", "AABB", "BVHNode", "IndexedTriangle"):
        assert re.search(rf"impl FlattenObject for {t}\b", geo), t
    for t in ("Lambertian", "Metal", "Dielectric", "ParameterizedMaterial", "Isotropic"):
        assert re.search(rf"impl FlattenMaterial for {t}\b", mat), t
    assert "impl FlattenObject for Scene" in tra and "pub fn render_to_image(&self) -> RgbImage" in tra
    assert re.search(r"impl Camera \{\s*pub fn flatten\(&self\) -> mi_rt::mi_camera_desc", tra)
    assert "pub fn flatten(&self, out: &mut super::mi_rt::SceneBuilder) -> i32" in tex and "to_rgb8()" in tex
    # struct literals: `mi_xxx { a: .., b: .. }` must name every field of mi_xxx exactly once (no `..` rest except Default ones)
    structs = _rust_structs()
    for name, src in files.items():
        code = _strip_rust(src)
        for m in re.finditer(r"\b(?:mi_rt::)?(mi_[a-z_]+)\s*\{", code):
            sname = m.group(1)
            before = code[max(0, m.start() - 40):m.start()]
            if sname not in structs or re.search(r"(struct|for|->)\s*$", before) or re.search(r"(pub\s+struct|impl)\s+$", before):
                continue
            depth, j = 1, m.end()
            while depth:
                depth += {"{": 1, "}": -1}.get(code[j], 0)
                j += 1
            body = code[m.end():j - 1]
            top, d = "", 0
            for ch in body:
                d += ch in "([{"
                d -= ch in ")]}"
                top += ch if d == 0 or (d == 1 and ch in "([{") else " "
            names = re.findall(r"(?:^|,)\s*(\w+)\s*:", top)
            want = [f for f, _ in structs[sname]]
            if ".." in top:
                assert set(names) <= set(want), (name, sname, names)
            else:
                assert sorted(names) == sorted(want), (name, sname, sorted(names), sorted(want))


def test_reference_patch_applies(tmp_path):
    """rust/reference.patch (five one-line edits + four include! lines) applies to the reference's sources as they are."""
    ref = "/root/reference/src"
    if not os.path.isdir(ref):
        pytest.skip("the reference is not on this box")
    import shutil
    shutil.copytree(ref, tmp_path / "src")
    out = subprocess.run(["patch", "-p1", "--dry-run", "-i", os.path.join(ROOT, "rust", "reference.patch")], cwd=tmp_path,
                         capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr
    patch = open(os.path.join(ROOT, "rust", "reference.patch")).read()
    for frag in ("geometry_flatten.rs", "materials_flatten.rs", "texture_flatten.rs", "tracing_flatten.rs"):
        assert f'include!("{frag}");' in patch
    assert "pub mod mi_rt;" in patch and "FlattenObject" in patch and "FlattenMaterial" in patch and "render_to_image_cpu" in patch
