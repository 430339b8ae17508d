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


def test_option_table_parses_names_and_values(tmp_path):
    """metabuli_amd/csrc/mtb_options.h (the library's experiment switches, read from the environment once per context): every name of the table is
    settable, flags follow "the variable exists", integers and the join variant parse, unknown names and bad variants are refused -- and no
    source of the product library calls getenv outside that header."""
    import subprocess
    src = tmp_path / "t.cpp"
    src.write_text(r'''
#include "mtb_options.h"
#include <cstdio>
int main() {
    MtbOptions o;
    int bad = 0;
    for (const mtbopt::Entry &e : mtbopt::kTable) bad += !mtbopt::set(&o, e.name, e.kind == mtbopt::VARIANT ? "q2w5" : "1");
    bad += !(o.join_variant == 0x25 && o.no_score_many == 1 && o.dir_depth == 1 && o.open_chunk == 1 && o.segm_clear[0] == '1');
    bad += !mtbopt::set(&o, "MTB_JOIN_VARIANT", "window") || o.join_variant != 0x100;
    bad += !mtbopt::set(&o, "MTB_JOIN_VARIANT", "auto") || o.join_variant != 0;
    bad += mtbopt::set(&o, "MTB_JOIN_VARIANT", "q3w9");                 /* refused, value kept */
    bad += mtbopt::set(&o, "MTB_NO_SUCH_SWITCH", "1");
    bad += !mtbopt::set(&o, "MTB_NO_SCORE_MANY", nullptr) || o.no_score_many != 0;      /* unset */
    bad += !mtbopt::set(&o, "MTB_NO_SCORE_MANY", "0") || o.no_score_many != 1;          /* a flag is on when the variable exists */
    bad += !mtbopt::set(&o, "MTB_JOIN_WIN", nullptr) || o.join_win != -1;               /* back to the default */
    bad += !mtbopt::set(&o, "MTB_OPEN_CHUNK", "4096") || o.open_chunk != 4096;
    setenv("MTB_DIR_DEPTH", "7", 1); setenv("MTB_JOIN_WIN_QT", "64", 1);
    MtbOptions e; mtbopt::from_environment(&e);
    bad += !(e.dir_depth == 7 && e.join_win_qt == 64 && e.join_win == -1);
    printf("bad %d\n", bad);
    return bad;
}
''')
    exe = tmp_path / "t"
    csrc = os.path.join(ROOT, "metabuli_amd", "csrc")
    subprocess.check_call(["g++", "-std=c++17", "-Wall", "-Werror", "-I", csrc, "-o", str(exe), str(src)])
    env = {k: v for k, v in os.environ.items() if not k.startswith("MTB_")}
    assert subprocess.run([str(exe)], env=env).returncode == 0
    for f in os.listdir(csrc):
        if f.endswith((".h", ".hip")) and f != "mtb_options.h":
            assert "getenv" not in open(os.path.join(csrc, f)).read(), f


def test_share_record_size_matches_the_binding(tmp_path):
    """mtb_index_share (the record a resident index is handed to other processes with) is plain old data of the size the Python binding allocates"""
    import subprocess
    import metabuli_amd as M
    src = tmp_path / "s.c"
    src.write_text('#include <stdio.h>\n#include "mtb.h"\nint main(void) { printf("%zu\\n", sizeof(mtb_index_share)); return 0; }\n')
    exe = tmp_path / "s"
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), "-o", str(exe), str(src)])
    assert int(subprocess.check_output([str(exe)]).decode()) == M.SHARE_BYTES
