"""The three statements of `tmpc_problem` -- the C header, the product's ctypes mirror and the reference-side binding
INTEGRATION.md shows a maintainer -- must agree field for field (names, order, C types).  CPU only."""
import ctypes as C
import os
import re

import common
from LinearMPCOverNetworks import _native

ROOT = os.path.dirname(common.PKG)
CTYPE = {"int32_t": C.c_int32, "double": C.c_double, "const double *": C.POINTER(C.c_double)}


def header_fields():
    """[(name, ctype)] of `typedef struct tmpc_problem { ... }` in include/tmpc.h"""
    src = open(os.path.join(ROOT, "include", "tmpc.h")).read()
    body = re.search(r"typedef struct tmpc_problem \{(.*?)\} tmpc_problem;", src, re.S).group(1)
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    out = []
    for decl in body.split(";"):
        decl = " ".join(decl.split())
        if not decl:
            continue
        m = re.match(r"(const double|int32_t|double)\s+(.*)", decl)
        assert m, decl
        base, names = m.groups()
        for n in names.split(","):
            n = n.strip()
            ptr = n.startswith("*")
            n = n.lstrip("* ")
            out.append((n, CTYPE["const double *" if ptr else base]))
    return out


def integration_stub_fields():
    """the `_fields_` expression of INTEGRATION.md's `_TmpcProblem`, evaluated as written"""
    md = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    m = re.search(r"class _TmpcProblem\(C\.Structure\):[^\n]*\n\s*_fields_ = (\(.*?\n)\n", md, re.S)
    assert m, "INTEGRATION.md no longer shows the _TmpcProblem stub"
    expr = re.sub(r"#[^\n]*", "", m.group(1))
    return eval(expr, {"C": C})


def same(a, b):
    return [(n, t) for n, t in a] == [(n, t) for n, t in b]


def test_header_ctypes_and_integration_stub_agree():
    hdr = header_fields()
    assert [n for n, _ in hdr][-2:] == ["rTP", "terminal_equality"]
    prod = list(_native.TmpcProblem._fields_)
    assert [n for n, _ in hdr] == [n for n, _ in prod]
    for (n, th), (_, tp) in zip(hdr, prod):
        assert C.sizeof(th) == C.sizeof(tp) and (th is tp or issubclass(tp, C._Pointer) == issubclass(th, C._Pointer)), n
    stub = integration_stub_fields()
    assert [n for n, _ in stub] == [n for n, _ in hdr]
    for (n, th), (_, ts) in zip(hdr, stub):
        assert C.sizeof(th) == C.sizeof(ts), n

    class Stub(C.Structure):
        _fields_ = stub
    assert C.sizeof(Stub) == C.sizeof(_native.TmpcProblem)
    for n, _ in hdr:
        assert getattr(Stub, n).offset == getattr(_native.TmpcProblem, n).offset, n


def test_abi_version_matches_header():
    src = open(os.path.join(ROOT, "include", "tmpc.h")).read()
    v = int(re.search(r"#define\s+TMPC_ABI_VERSION\s+(\d+)", src).group(1))
    assert v == _native.ABI_VERSION == _native.lib().tmpc_abi_version()
