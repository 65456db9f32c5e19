"""integration/rust/sdf-gpu/src/lib.rs never meets a compiler in this image (no Rust toolchain), so a drifted
declaration would be found by its first user.  This is the check that CAN be made here: every function
include/sdfgpu.h declares is declared in lib.rs's `extern "C"` block with the same number of arguments, the
same pointer-ness and constness per argument and a matching return kind -- and lib.rs declares nothing the
header does not have."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _strip_c_comments(s):
    return re.sub(r"/\*.*?\*/", " ", s, flags=re.S)


def _split_args(a):
    a = a.strip()
    if not a or a == "void":
        return []
    return [x.strip() for x in a.split(",")]


def header_functions():
    src = _strip_c_comments(open(os.path.join(ROOT, "include", "sdfgpu.h")).read())
    src = "\n".join(l for l in src.splitlines() if not l.lstrip().startswith("#"))
    out = {}
    for m in re.finditer(r"([A-Za-z_][\w\s\*]*?)\b(sdfgpu_\w+)\s*\(([^()]*)\)\s*;", src):
        ret, name, args = m.group(1).strip(), m.group(2), _split_args(m.group(3))
        if "typedef" in ret:
            continue
        kinds = []
        for a in args:
            stars = a.count("*") + a.count("[")
            # constness of what the OUTERMOST pointer points at: the qualifier written just before the last '*'
            # (`sdfgpu_ctx* const* ctxs` = pointer to const pointer = Rust's `*const *mut Ctx`)
            const = "const" in a.split("*")[-2] if "*" in a else ("const" in a and "[" in a)
            kinds.append((stars, bool(const) if stars else False))
        out[name] = ("void" if ret == "void" else ("ptr" if "*" in ret else "val"), kinds)
    return out


def rust_functions():
    src = open(os.path.join(ROOT, "integration", "rust", "sdf-gpu", "src", "lib.rs")).read()
    src = re.sub(r"//[^\n]*", "", src)
    block = src[src.index('extern "C" {'):]
    block = block[:block.index("\n}\n")]
    out = {}
    for m in re.finditer(r"\bfn\s+(sdfgpu_\w+)\s*\(([^()]*)\)\s*(->\s*[^;]+)?;", block):
        name, args, ret = m.group(1), _split_args(m.group(2)), (m.group(3) or "").replace("->", "").strip()
        kinds = []
        for a in args:
            ty = a.split(":", 1)[1].strip()
            stars = ty.count("*")
            const = ty.startswith("*const") if stars else False
            kinds.append((stars, const))
        out[name] = ("void" if not ret else ("ptr" if ret.startswith("*") else "val"), kinds)
    return out


def test_every_header_function_is_bound_with_the_same_shape():
    h, r = header_functions(), rust_functions()
    assert len(h) >= 35, sorted(h)                      # the parser really sees the header
    assert sorted(set(h) - set(r)) == [], "declared in sdfgpu.h but not in lib.rs"
    assert sorted(set(r) - set(h)) == [], "declared in lib.rs but not in sdfgpu.h"
    for name in sorted(h):
        hret, hk = h[name]
        rret, rk = r[name]
        assert hret == rret, f"{name}: return kind {hret} (C) vs {rret} (Rust)"
        assert len(hk) == len(rk), f"{name}: {len(hk)} arguments in sdfgpu.h, {len(rk)} in lib.rs"
        for i, (ha, ra) in enumerate(zip(hk, rk)):
            assert ha[0] == ra[0], f"{name}: argument {i} pointer depth {ha[0]} (C) vs {ra[0]} (Rust)"
            if ha[0]:
                assert ha[1] == ra[1], f"{name}: argument {i} constness differs (C const={ha[1]}, Rust const={ra[1]})"


def test_constants_agree():
    hdr = open(os.path.join(ROOT, "include", "sdfgpu.h")).read()
    rs = open(os.path.join(ROOT, "integration", "rust", "sdf-gpu", "src", "lib.rs")).read()
    pairs = {"SDFGPU_COMM_ID_BYTES": "COMM_ID_BYTES", "SDFGPU_STATIC_MAX_POINTS": "STATIC_MAX_POINTS"}
    for c_name, r_name in pairs.items():
        cv = int(re.search(rf"#define\s+{c_name}\s+(\d+)", hdr).group(1))
        rv = int(re.search(rf"pub const {r_name}: \w+ = ([\d_]+);", rs).group(1).replace("_", ""))
        assert cv == rv, (c_name, cv, rv)
    for c_name, r_name in (("SDFGPU_OPT_REORDER", "OPT_REORDER"), ("SDFGPU_OPT_SKIP_EXTENSION", "OPT_SKIP_EXTENSION"),
                           ("SDFGPU_OPT_COMPACTION", "OPT_COMPACTION"), ("SDFGPU_OPT_TILE_ORDER", "OPT_TILE_ORDER"),
                           ("SDFGPU_OPT_STATS_ASM_WALK", "OPT_STATS_ASM_WALK"), ("SDFGPU_OPT_QUERY_PIPELINE", "OPT_QUERY_PIPELINE"),
                           ("SDFGPU_EXT_SMOOTH_UNION", "EXT_SMOOTH_UNION"), ("SDFGPU_EXT_BOX_FRAME", "EXT_BOX_FRAME"),
                           ("SDFGPU_EXT_CLONE_REPAIR", "EXT_CLONE_REPAIR")):
        cv = int(re.search(rf"{c_name}\s*=\s*(\d+)", hdr).group(1))
        rv = int(re.search(rf"pub const {r_name}: \w+ = (\d+);", rs).group(1))
        assert cv == rv, (c_name, cv, rv)
