"""The C-ABI library loads and exports every function include/mtb.h declares
(no compute calls here: there is no GPU in the CPU test run)."""
import ctypes as C
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    src = open(os.path.join(ROOT, "include", "mtb.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(mtb_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    import metabuli_amd as M
    if not os.path.exists(M.LIB_PATH):
        M.build()
    lib = C.CDLL(M.LIB_PATH)
    names = _declared()
    assert len(names) >= 25
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, missing


def test_structs_have_the_documented_sizes():
    import metabuli_amd as M
    assert M.kmer_dt.itemsize == 16 and M.match_dt.itemsize == 24 and M.result_dt.itemsize == 24
    assert C.sizeof(M.Params) == 44


def test_no_gpu_means_a_loud_error_not_a_fallback():
    import metabuli_amd as M
    try:
        import torch
        if torch.cuda.is_available():
            return
    except Exception:
        pass
    try:
        M.Context(0)
    except M.MtbError as e:
        assert e.status == M.MTB_ERR_DEVICE
    else:
        raise AssertionError("Context creation must fail without a GPU")


def test_headers_compile_as_c99_and_cxx17(tmp_path):
    """include/mtb.h is a plain C header (the drop-in boundary), include/mtb.hpp a header-only C++17 shim over it"""
    import subprocess
    c = tmp_path / "t.c"; c.write_text('#include "mtb.h"\nint main(void) { mtb_params p; mtb_default_params(&p); return (int)sizeof(mtb_match) - 24; }\n')
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Werror", "-fsyntax-only", "-I", os.path.join(ROOT, "include"), str(c)])
    cc = tmp_path / "t.cpp"; cc.write_text('#include "mtb.hpp"\nint main() { return sizeof(mtb::Kmer) == 16 ? 0 : 1; }\n')
    subprocess.check_call(["g++", "-std=c++17", "-fsyntax-only", "-I", os.path.join(ROOT, "include"), str(cc)])
