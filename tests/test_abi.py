"""CPU tests of the drop-in boundary: the C-ABI library loads and exports every symbol
include/mi355_cholmod.h declares; without a GPU it fails loudly instead of falling back."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from conftest import ROOT, load_golden, sym_of


def declared_symbols(header="mi355_cholmod.h"):
    txt = open(os.path.join(ROOT, "include", header)).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(mi355_[a-z_A-Z0-9]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol():
    from suitesparse_amd import engine
    L = engine.lib()
    syms = declared_symbols()
    assert len(syms) >= 14
    for s in syms:
        assert hasattr(L, s), f"{s} declared in include/mi355_cholmod.h but not exported"
    assert set(syms) == set(engine.EXPORTS)
    from suitesparse_amd import symbolic
    syms2 = declared_symbols("mi355_symbolic.h")
    for s in syms2:
        assert hasattr(L, s), f"{s} declared in include/mi355_symbolic.h but not exported"
    assert set(syms2) == set(symbolic.EXPORTS)


def test_no_cpu_fallback_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from suitesparse_amd import engine
    assert engine.probe() == 0
    g = load_golden("demo_bcsstk01")
    with pytest.raises(engine.Mi355Error) as ei:
        engine.SupernodalFactor(sym_of(g))
    assert ei.value.code == engine.CHOLMOD_GPU_PROBLEM


def test_product_does_not_link_oracle():
    """the product library and package must not reference anything under oracle/"""
    for root, _, files in os.walk(os.path.join(ROOT, "suitesparse_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".cpp", ".h")):
                txt = open(os.path.join(root, f)).read()
                assert "oracle" not in txt.replace("the oracle", ""), f"{f} mentions oracle/"


def test_product_knows_nothing_of_the_cpu_emulation():
    """tests/helpers/hipcpu and libmi355_cpuemu.so are test infrastructure: nothing under suitesparse_amd/, bench.py
    or __graft_entry__.smoke() may name them (only build() compiles them, and only tests load them)"""
    import inspect
    for root, _, files in os.walk(os.path.join(ROOT, "suitesparse_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".cpp", ".h")):
                txt = open(os.path.join(root, f)).read()
                assert "cpuemu" not in txt and "hipcpu" not in txt, f
    assert "cpuemu" not in open(os.path.join(ROOT, "bench.py")).read()
    import __graft_entry__ as ge
    assert "cpuemu" not in inspect.getsource(ge.smoke)
    from suitesparse_amd import engine
    assert engine.LIB_PATH.endswith(os.path.join("suitesparse_amd", "libmi355_cholmod.so"))


def test_cholmod_side_glue_exports_its_header():
    """include/cholmod_mi355.h (the CHOLMOD side of the drop-in) against the patched build of the reference: every
    declared hook is defined there, and every mi355_* symbol that build leaves undefined is declared in
    include/mi355_cholmod.h (so it binds to the engine and to nothing else)"""
    import subprocess
    so = os.path.join(ROOT, "oracle", "_ref", "libcholmod_mi355.so")
    if not os.path.exists(so):
        if os.path.isdir("/root/reference/CHOLMOD"):
            pytest.fail("oracle/_ref/libcholmod_mi355.so missing: run __graft_entry__.build()")
        pytest.skip("no patched reference build here")
    txt = open(os.path.join(ROOT, "include", "cholmod_mi355.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    hooks = sorted(set(re.findall(r"\b(cholmod_mi355_[a-z0-9_]+)\s*\(", txt)))
    assert len(hooks) == 11
    nm = subprocess.run(["nm", "-D", so], capture_output=True, text=True, check=True).stdout.splitlines()
    defined = {ln.split()[-1] for ln in nm if " T " in ln}
    undefined = {ln.split()[-1] for ln in nm if " U " in ln and ln.split()[-1].startswith("mi355_")}
    for h in hooks:
        assert h in defined, h
    assert undefined and undefined <= set(declared_symbols()), undefined - set(declared_symbols())


def test_rccl_declarations_match_the_installed_header():
    """multi.hip.hpp declares the slice of rccl.h it binds with dlsym by hand; the installed header must agree (enum
    values, the 128-byte id passed by value, every signature), or the first run on several GPUs would be the check."""
    import re
    import shutil
    import subprocess
    hdr = "/opt/rocm/include/rccl/rccl.h"
    if not os.path.exists(hdr) or shutil.which("g++") is None:
        pytest.skip("no rccl.h / g++ on this machine")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = open(os.path.join(root, "suitesparse_amd", "csrc", "multi.hip.hpp")).read()
    m = re.search(r"constexpr int RCCL_F64 = (\d+), RCCL_I64 = (\d+), RCCL_SUM = (\d+), RCCL_MIN = (\d+);", src)
    assert m, "the constants of multi.hip.hpp moved"
    assert "struct RcclId { char internal[128]; };" in src
    # the pointer types the engine declares, spelled with ints and void * (what the check file spells with rccl's types)
    for decl in ["int (*GetUniqueId)(RcclId *)", "int (*CommInitRank)(void **, int, RcclId, int)", "int (*CommDestroy)(void *)",
                 "int (*CommSplit)(void *, int, int, void **, void *)",
                 "int (*Reduce)(const void *, void *, size_t, int, int, int, void *, hipStream_t)",
                 "int (*Broadcast)(const void *, void *, size_t, int, int, void *, hipStream_t)",
                 "int (*AllReduce)(const void *, void *, size_t, int, int, void *, hipStream_t)",
                 "int (*GroupStart)()", "int (*GroupEnd)()", "const char *(*GetErrorString)(int)"]:
        assert decl in src, decl
    f64, i64, s_, mn = m.groups()
    r = subprocess.run(["g++", "-std=c++17", "-fsyntax-only", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include",
                        f"-DF64={f64}", f"-DI64={i64}", f"-DSUM={s_}", f"-DMIN={mn}",
                        os.path.join(root, "tests", "helpers", "rccl_abi_check.cpp")], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
