"""CPU: the C-ABI library loads and exports every symbol include/fastmatch_hip.h declares;
without a GPU the product path fails loudly instead of falling back."""
import os
import re

import pytest

import fastmatch_amd
from fastmatch_amd import _ffi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    hdr = open(os.path.join(ROOT, "include", "fastmatch_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    return sorted(set(re.findall(r"\b(fm_[a-z0-9_]+)\s*\(", hdr)))


def test_header_symbols_all_exported_and_bound():
    names = _declared()
    assert len(names) >= 15
    lib = _ffi.load_library()
    for n in names:
        assert hasattr(lib, n), "libfastmatch_hip.so does not export %s" % n
        assert n in _ffi.SYMBOLS, "ctypes binding lacks %s" % n
    assert sorted(_ffi.SYMBOLS) == names, "binding declares symbols the header does not"


def test_no_cpu_fallback_without_device():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(fastmatch_amd.FastMatchHipError) as e:
        fastmatch_amd.Context(0)
    assert "no CPU fallback" in str(e.value)
    from fastmatch_amd import matchutil
    import numpy as np
    with pytest.raises(fastmatch_amd.FastMatchHipError):
        matchutil.bf_match(np.zeros((2, 128), np.uint8), np.zeros((2, 128), np.uint8), k=2)


def test_product_does_not_import_oracle():
    pkg = os.path.join(ROOT, "fast-match_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(import|from)\s+oracle\b", src, flags=re.M), f
                assert "liboracle" not in src, f


def test_abi_revision_of_header_library_and_binding_agree():
    """fm_abi_version() (a signature changed without a version to check): the header's FM_ABI_VERSION, what the
    built library returns and what the ctypes binding was written for are one number, and the size-checked statistics
    struct of the binding has the header's layout."""
    hdr = open(os.path.join(ROOT, "include", "fastmatch_hip.h")).read()
    ver = int(re.search(r"#define\s+FM_ABI_VERSION\s+(\d+)", hdr).group(1))
    lib = _ffi.load_library()
    assert lib.fm_abi_version() == ver == _ffi.FM_ABI_VERSION
    fields = re.search(r"typedef struct fm_stats_ex \{(.*?)\} fm_stats_ex;", hdr, flags=re.S).group(1)
    names = re.findall(r"\b([a-z_]+)\s*[,;]", re.sub(r"/\*.*?\*/", "", fields, flags=re.S))
    assert names == [f[0] for f in _ffi.fm_stats_ex._fields_]


def test_struct_layouts_of_the_binding_match_the_header_as_a_c_compiler_sees_it(tmp_path):
    """The ctypes mirrors of fm_stats, fm_stats_ex and fm_expand_desc against sizeof / offsetof from gcc on the header:
    same size, every field at the same offset, in the same order."""
    import ctypes
    import subprocess
    structs = {"fm_stats": _ffi.fm_stats, "fm_stats_ex": _ffi.fm_stats_ex, "fm_expand_desc": _ffi.fm_expand_desc}
    lines = ['#include <stdio.h>', '#include <stddef.h>', '#include "fastmatch_hip.h"', 'int main(void) {']
    for name, cls in structs.items():
        lines.append('printf("%s size %%zu\\n", sizeof(%s));' % (name, name))
        for f in cls._fields_:
            lines.append('printf("%s %s %%zu\\n", offsetof(%s, %s));' % (name, f[0], name, f[0]))
    lines += ['return 0; }']
    src = tmp_path / "layout.c"
    src.write_text("\n".join(lines))
    exe = str(tmp_path / "layout")
    subprocess.check_call(["gcc", "-std=c99", "-I", os.path.join(ROOT, "include"), str(src), "-o", exe])
    out = subprocess.run([exe], capture_output=True, text=True, check=True).stdout.split("\n")
    seen = 0
    for line in out:
        if not line:
            continue
        name, field, value = line.split()
        cls = structs[name]
        if field == "size":
            assert ctypes.sizeof(cls) == int(value), name
        else:
            assert getattr(cls, field).offset == int(value), (name, field)
        seen += 1
    assert seen == sum(len(c._fields_) + 1 for c in structs.values())
    # ... and the header declares no field the binding lacks
    hdr = re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", "fastmatch_hip.h")).read(), flags=re.S)
    body = re.search(r"typedef struct fm_expand_desc \{(.*?)\} fm_expand_desc;", hdr, flags=re.S).group(1)
    names = re.findall(r"\b([a-z_0-9]+)\s*[,;]", body)
    assert names == [f[0] for f in _ffi.fm_expand_desc._fields_]


def test_every_prototype_of_the_header_matches_the_binding_argument_by_argument():
    """ADVICE r03 found a signature that had changed under the binding.  Every prototype of the header against the ctypes
    argtypes: same number of parameters, and per parameter the same class (pointer, 64-bit integer, 32-bit integer, double,
    float) -- an inserted or reordered argument cannot go unnoticed."""
    import ctypes
    hdr = re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", "fastmatch_hip.h")).read(), flags=re.S)
    protos = re.findall(r"\b(?:const\s+char\s*\*|int|void)\s+(fm_[a-z0-9_]+)\s*\(([^;{]*?)\)\s*;", hdr, flags=re.S)
    assert len(protos) == len(_ffi.SYMBOLS)

    def klass_of_text(p):
        p = " ".join(p.split())
        if "*" in p or "[" in p:
            return "ptr"
        if re.search(r"\bdouble\b", p):
            return "f64"
        if re.search(r"\bfloat\b", p):
            return "f32"
        if re.search(r"\b(int64_t|uint64_t|size_t|long long)\b", p):
            return "i64"
        if re.search(r"\b(int32_t|uint32_t|int|unsigned)\b", p):
            return "i32"
        raise AssertionError("unclassified parameter: " + p)

    def klass_of_ctype(t):
        if t in (ctypes.c_void_p, ctypes.c_char_p) or hasattr(t, "contents") or getattr(t, "_type_", None) is not None and issubclass(t, ctypes._Pointer):
            return "ptr"
        return {ctypes.c_double: "f64", ctypes.c_float: "f32", ctypes.c_int64: "i64", ctypes.c_uint64: "i64",
                ctypes.c_int: "i32", ctypes.c_int32: "i32", ctypes.c_uint32: "i32"}[t]

    for name, params in protos:
        params = params.strip()
        texts = [] if params in ("", "void") else [p for p in params.split(",")]
        argtypes = _ffi.SYMBOLS[name][1]
        assert len(texts) == len(argtypes), "%s: header has %d parameters, the binding %d" % (name, len(texts), len(argtypes))
        for i, (p, t) in enumerate(zip(texts, argtypes)):
            assert klass_of_text(p) == klass_of_ctype(t), "%s: parameter %d (%s) is bound as %s" % (name, i, " ".join(p.split()), t)
