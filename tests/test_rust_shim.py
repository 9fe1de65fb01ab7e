"""The Rust shim (nbody-llm_amd/rust/src/lib.rs) has never met a compiler here (no rustc in the image; the reference needs
nightly and network access).  What a missing compiler still allows: its `unsafe extern "C"` block and its #[repr(C)] structs
are parsed and compared with include/nbody_hip.h -- names, arity, pointer/scalar types, return types, field order -- the way
tests/test_abi.py checks the ctypes mirror; and the traits it implements are compared with the reference's definitions
where /root/reference is present (this container; the GPU box has no reference and skips that part)."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "nbody_hip.h")
LIB_RS = os.path.join(ROOT, "nbody-llm_amd", "rust", "src", "lib.rs")
REFERENCE = "/root/reference"

C_SCALARS = {"int": "i32", "int32_t": "i32", "uint32_t": "u32", "uint64_t": "u64", "size_t": "usize", "float": "f32", "double": "f64",
             "void": "void", "char": "c_char", "long long": "i64", "unsigned long long": "u64"}
RUST_SCALARS = {"c_int": "i32", "i32": "i32", "u32": "u32", "u64": "u64", "usize": "usize", "f32": "f32", "f64": "f64", "c_void": "void",
                "c_char": "c_char", "i64": "i64"}


def strip_c_comments(text):
    return re.sub(r"/\*.*?\*/", "", text, flags=re.S)


def canon_c(t):
    """'const float center[3]' / 'NbodyHandle** out' / 'size_t n' -> canonical type string (argument names dropped)"""
    t = t.strip()
    arr = bool(re.search(r"\[\d*\]\s*$", t))
    t = re.sub(r"\[\d*\]\s*$", "", t).strip()
    stars = t.count("*")
    t = t.replace("*", " ")
    toks = t.split()
    const = "const" in toks
    toks = [x for x in toks if x != "const"]
    # drop the argument name: the last token unless the type is a single token (e.g. 'void')
    known = set(C_SCALARS) | {"NbodyHandle", "NbodyConfig", "NbodyStats", "NbodyLetStats", "unsigned", "long"}
    if len(toks) > 1 and toks[-1] not in known:
        toks = toks[:-1]
    base = " ".join(toks)
    base = C_SCALARS.get(base, base)
    depth = stars + (1 if arr else 0)
    out = base
    for level in range(depth):
        # constness applies to the pointee of the innermost pointer in every prototype of this header
        out = ("*const " if (const and level == 0) else "*mut ") + out
    return out


def canon_rust(t):
    t = t.strip()
    m = re.match(r"\*(const|mut)\s+(.*)", t)
    if m:
        return f"*{m.group(1)} {canon_rust(m.group(2))}"
    return RUST_SCALARS.get(t, t)


def header_prototypes():
    text = strip_c_comments(open(HEADER).read())
    protos = {}
    for ret, name, args in re.findall(r"^\s*((?:const\s+)?\w[\w\s]*?\*?)\s*\b(nbody_\w+)\s*\(([^)]*)\)\s*;", text, flags=re.M):
        arglist = [] if args.strip() in ("", "void") else [canon_c(a) for a in args.split(",")]
        protos[name] = (canon_c(ret + " x") if "*" in ret else C_SCALARS.get(ret.strip(), ret.strip()), arglist)
    return protos


def rust_externs():
    text = open(LIB_RS).read()
    block = re.search(r'unsafe extern "C" \{(.*?)\n\}', text, flags=re.S).group(1)
    fns = {}
    for name, args, ret in re.findall(r"fn\s+(nbody_\w+)\s*\(([^)]*)\)\s*(?:->\s*([^;]+))?;", block):
        arglist = [canon_rust(a.split(":", 1)[1]) for a in args.split(",") if a.strip()]
        fns[name] = (canon_rust(ret) if ret else "void", arglist)
    return fns


def test_every_extern_fn_matches_its_prototype_in_the_header():
    protos, fns = header_prototypes(), rust_externs()
    assert len(fns) >= 20 and len(protos) >= 50
    for name, (ret, args) in fns.items():
        assert name in protos, f"{name} is not declared in include/nbody_hip.h"
        cret, cargs = protos[name]
        assert len(args) == len(cargs), f"{name}: {len(args)} arguments in lib.rs, {len(cargs)} in the header"
        assert args == cargs, f"{name}: lib.rs {args} vs header {cargs}"
        assert ret == cret, f"{name}: returns {ret} in lib.rs, {cret} in the header"


def test_the_trait_surface_is_bound():
    """every entry point the trait methods need is in the extern block"""
    need = {"nbody_create", "nbody_destroy", "nbody_clone", "nbody_upload", "nbody_download", "nbody_count", "nbody_add_point", "nbody_remove_point",
            "nbody_set_settings", "nbody_set_settings_f64", "nbody_set_bounds", "nbody_set_bounds_f64", "nbody_init", "nbody_step_by", "nbody_step_by_f64",
            "nbody_steps", "nbody_update_forces", "nbody_sync", "nbody_last_error", "nbody_tree_export_cells"}
    assert need <= set(rust_externs())


def c_struct_fields(name):
    text = strip_c_comments(open(HEADER).read())
    body = re.search(r"typedef struct %s \{(.*?)\} %s;" % (name, name), text, flags=re.S).group(1)
    fields = []
    for decl in body.split(";"):
        decl = decl.strip()
        if not decl:
            continue
        m = re.match(r"([\w\s]+?)\s+(\w+)(\[\d+\])?$", decl)
        fields.append((m.group(2), C_SCALARS.get(m.group(1).strip(), m.group(1).strip()) + (m.group(3) or "")))
    return fields


def rust_struct_fields(name):
    text = open(LIB_RS).read()
    m = re.search(r"#\[repr\(C\)\]\s*(?:pub\s+)?struct %s \{(.*?)\n\}" % name, text, flags=re.S)
    body = re.sub(r"//[^\n]*", "", m.group(1))
    return [(f.split(":")[0].strip().replace("pub ", ""), canon_rust(f.split(":")[1])) for f in body.split(",") if ":" in f]


def test_nbody_config_has_the_headers_fields_in_the_headers_order():
    assert rust_struct_fields("NbodyConfig") == c_struct_fields("NbodyConfig")
    assert len(c_struct_fields("NbodyConfig")) == 13


def test_constants_agree_with_the_header():
    text = strip_c_comments(open(HEADER).read())
    rs = open(LIB_RS).read()
    enums = dict((k, int(v)) for k, v in re.findall(r"\b(NBODY_[A-Z0-9_]+)\s*=\s*(-?\d+)", text))
    for rust_name, c_name in (("NBODY_BRUTE_FORCE", "NBODY_BRUTE_FORCE"), ("NBODY_BARNES_HUT", "NBODY_BARNES_HUT")):
        assert int(re.search(r"pub const %s: i32 = (\d+);" % rust_name, rs).group(1)) == enums[c_name]
    variants = {"MathMode": {"Strict": "NBODY_MATH_STRICT", "Fast": "NBODY_MATH_FAST"},
                "LeafMode": {"Reference": "NBODY_LEAF_REFERENCE", "Direct": "NBODY_LEAF_DIRECT"},
                "TreeBuild": {"Host": "NBODY_TREE_HOST", "Device": "NBODY_TREE_DEVICE", "Auto": "NBODY_TREE_AUTO"}}
    for enum, table in variants.items():
        body = re.sub(r"//[^\n]*", "", re.search(r"pub enum %s \{(.*?)\n\}" % enum, rs, flags=re.S).group(1))
        got = dict((k, int(v)) for k, v in re.findall(r"(\w+)\s*=\s*(\d+)", body))
        assert got == {k: enums[c] for k, c in table.items()}, enum
    for dtype, c_name in (("f32", "NBODY_F32"), ("f64", "NBODY_F64")):
        m = re.search(r"impl HipFloat for %s \{\s*const DTYPE: i32 = (\d+);" % dtype, rs)
        assert int(m.group(1)) == enums[c_name]


@pytest.mark.skipif(not os.path.isdir(REFERENCE), reason="the reference's sources are only in the build container")
def test_the_implemented_traits_have_the_references_methods():
    """`impl Simulation<..> for HipSimulation` defines exactly the methods the reference's trait requires (src/shared.rs:80-97;
    `step` has a default body there), and `impl Renderable` those of src/render/mod.rs:12-15."""
    shared = open(os.path.join(REFERENCE, "src", "shared.rs")).read()
    trait = re.search(r"pub trait Simulation<[^{]*\{(.*?)\n\}", shared, flags=re.S).group(1)
    required = set(re.findall(r"fn (\w+)\s*\([^)]*\)[^;{]*;", trait))
    defaulted = set(re.findall(r"fn (\w+)\s*\([^)]*\)[^;{]*\{", trait))
    rs = open(LIB_RS).read()
    impl = rs[rs.index("impl<F: HipFloat, const METHOD: i32> Simulation<F, 3, P<F>, I<F>> for HipSimulation<F, METHOD> {"):]
    impl = impl[:impl.index("\n}\n")]
    have = set(re.findall(r"\n    fn (\w+)\s*\(", impl))
    assert required <= have, required - have
    assert have <= required | defaulted, have - required - defaulted
    render = open(os.path.join(REFERENCE, "src", "render", "mod.rs")).read()
    rtrait = re.search(r"pub trait Renderable \{(.*?)\n\}", render, flags=re.S).group(1)
    rneed = set(re.findall(r"fn (\w+)", rtrait))
    rimpl = rs[rs.index("impl<F: HipFloat, const METHOD: i32> Renderable for HipSimulation<F, METHOD> {"):]
    assert set(re.findall(r"\n    fn (\w+)\s*\(", rimpl)) == rneed
