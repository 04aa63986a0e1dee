"""rustracer-hip-sys (the Rust FFI crate north_star names) against include/rustracer_hip.h.

There is no Rust toolchain in the build image, so the crate ships as source and this test is its compiler: it parses the header and
src/lib.rs and fails on any difference in the set of functions, their order, argument counts, argument and return kinds, and on any
struct whose repr(C) size (computed from the Rust field list) differs from the C compiler's sizeof."""
import os
import re
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "rustracer_hip.h")
LIB_RS = os.path.join(ROOT, "rustracer-hip-sys", "src", "lib.rs")

C_TO_RUST = {"int": "c_int", "void": "c_void", "char": "c_char", "float": "f32", "double": "f64", "uint8_t": "u8", "uint32_t": "u32", "uint64_t": "u64"}


def _strip_comments(text):
    return re.sub(r"/\*.*?\*/", " ", text, flags=re.S)


def _c_type(t):
    """C type text -> the Rust spelling lib.rs must use."""
    t = t.replace("volatile", " ").strip()
    stars = t.count("*")
    t = t.replace("*", " ")
    words = t.split()
    const = "const" in words
    base = [w for w in words if w != "const"]
    assert len(base) == 1, t
    r = C_TO_RUST.get(base[0], base[0])
    if stars == 0:
        return r
    out = ("*const " if const else "*mut ") + r
    for _ in range(stars - 1):
        out = "*mut " + out
    return out


def header_functions():
    text = _strip_comments(open(HEADER).read())
    body = text[text.index('extern "C" {'):]
    fns = []
    for m in re.finditer(r"([A-Za-z_][A-Za-z0-9_ \*]*?)\b(rt_[a-z0-9_]+)\s*\(([^()]*)\)\s*;", body):
        ret, name, args = m.group(1).strip(), m.group(2), m.group(3).strip()
        arglist = []
        if args and args != "void":
            for a in args.split(","):
                mm = re.match(r"(.*?)([A-Za-z_][A-Za-z0-9_]*)\s*$", a.strip(), flags=re.S)
                arglist.append((mm.group(2), _c_type(mm.group(1))))
        fns.append((name, None if ret == "void" else _c_type(ret), arglist))
    return fns


def rust_functions():
    text = re.sub(r"//[^\n]*", "", open(LIB_RS).read())
    ext = text[text.index('extern "C" {'):]
    ext = ext[:ext.index("\n}\n")]
    fns = []
    for m in re.finditer(r"pub fn (rt_[a-z0-9_]+)\s*\(([^()]*)\)\s*(?:->\s*([^;]+))?;", ext):
        name, args, ret = m.group(1), m.group(2).strip(), m.group(3)
        arglist = []
        for a in [x for x in args.split(",") if x.strip()]:
            n, t = a.split(":", 1)
            arglist.append((n.strip(), " ".join(t.split())))
        fns.append((name, None if ret is None else " ".join(ret.split()), arglist))
    return fns


def test_every_export_is_bound_with_the_same_signature():
    h, r = header_functions(), rust_functions()
    assert len(h) >= 35
    assert [f[0] for f in h] == [f[0] for f in r], "functions (and their order) differ: %s" % (set(f[0] for f in h) ^ set(f[0] for f in r))
    for (name, hret, hargs), (_, rret, rargs) in zip(h, r):
        assert hret == rret, (name, hret, rret)
        assert len(hargs) == len(rargs), (name, hargs, rargs)
        for (hn, ht), (rn, rt) in zip(hargs, rargs):
            assert ht == rt, "%s(%s): header %s, lib.rs %s" % (name, hn, ht, rt)
            assert hn == rn, "%s: argument named %s in the header, %s in lib.rs" % (name, hn, rn)


def test_the_library_exports_what_the_crate_links():
    from rustracer_amd import _lib
    names = [f[0] for f in rust_functions()]
    assert sorted(names) == sorted(_lib.EXPORTS)
    L = _lib.lib()
    assert [n for n in names if not hasattr(L, n)] == []


SIZES = {"u8": 1, "u32": 4, "f32": 4, "u64": 8, "f64": 8}


def _rust_struct_sizes():
    text = re.sub(r"//[^\n]*", "", open(LIB_RS).read())
    out = {}
    for m in re.finditer(r"#\[repr\(C\)\][^{]*?pub struct (rt_[a-z_]+)\s*\{([^}]*)\}", text, flags=re.S):
        off, align = 0, 1
        fields = [f.strip() for f in m.group(2).split(",") if ":" in f]
        if any("_private" in f for f in fields):
            continue
        for f in fields:
            t = f.split(":", 1)[1].strip()
            mm = re.match(r"\[(\w+);\s*(\d+)\]", t)
            base, count = (mm.group(1), int(mm.group(2))) if mm else (t, 1)
            sz = SIZES[base]
            off = (off + sz - 1) // sz * sz + sz * count
            align = max(align, sz)
        out[m.group(1)] = (off + align - 1) // align * align
    return out


def test_struct_sizes_equal_the_c_compilers(tmp_path):
    rust = _rust_struct_sizes()
    assert set(rust) == {"rt_material", "rt_render_params", "rt_scene_info", "rt_counters"}
    src = tmp_path / "sz.c"
    src.write_text('#include <stdio.h>\n#include "%s"\nint main(void){printf("%%zu %%zu %%zu %%zu\\n", sizeof(rt_material), sizeof(rt_render_params), '
                   'sizeof(rt_scene_info), sizeof(rt_counters)); return 0;}\n' % HEADER)
    exe = tmp_path / "sz"
    subprocess.check_call(["gcc", str(src), "-o", str(exe)])
    c = [int(v) for v in subprocess.check_output([str(exe)]).split()]
    assert [rust["rt_material"], rust["rt_render_params"], rust["rt_scene_info"], rust["rt_counters"]] == c
    # and the sizes the crate's own #[test] asserts
    text = open(LIB_RS).read()
    for name, size in zip(["rt_material", "rt_render_params", "rt_scene_info", "rt_counters"], c):
        assert "size_of::<%s>(), %d)" % (name, size) in text
