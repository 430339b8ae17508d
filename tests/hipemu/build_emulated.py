"""TEST INFRASTRUCTURE ONLY: builds the library's sources against the emulator header of this directory (see hip/hip_runtime.h) --
libmtb_hipemu.so + the driver linked against it -- into a directory OUTSIDE the package.  Usage:

    python tests/hipemu/build_emulated.py /tmp/mtb_hipemu [asan]
    MTB_HIPEMU=1 MTB_LIB=/tmp/mtb_hipemu/libmtb_hipemu.so python -m pytest tests -m gpu -q          # the GPU parity tests, executed by the emulator
"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
HERE = os.path.join(ROOT, "tests", "hipemu")


def build(out_dir, asan=False, opt="-O1"):
    os.makedirs(out_dir, exist_ok=True)
    lib = os.path.join(out_dir, "libmtb_hipemu.so")
    src = os.path.join(ROOT, "metabuli_amd", "csrc", "mtb_api.hip")
    deps = [src, os.path.join(HERE, "hip", "hip_runtime.h"), os.path.join(HERE, "hipemu_dyn_shared.h")] + \
           [os.path.join(ROOT, "metabuli_amd", "csrc", f) for f in os.listdir(os.path.join(ROOT, "metabuli_amd", "csrc")) if f.endswith(".h")]
    if not os.path.exists(lib) or os.path.getmtime(lib) < max(os.path.getmtime(d) for d in deps):
        # -fno-extern-tls-init: `extern __shared__ T s_dyn[]` is an extern thread_local here; without the flag g++ calls a TLS init wrapper through a
        # weak symbol that a shared object resolves to 0
        cmd = ["g++", "-x", "c++", "-std=c++17", opt, "-g", "-fPIC", "-shared", "-pthread", "-fno-extern-tls-init", "-I", HERE, "-include", os.path.join(HERE, "hipemu_dyn_shared.h"),
               "-Wno-unknown-pragmas", "-Wno-attributes", "-o", lib, src]
        if asan:
            cmd[5:5] = ["-fsanitize=address", "-fno-omit-frame-pointer"]
        cmd[5:5] = os.environ.get("HIPEMU_CXXFLAGS", "").split()          # tuning variants (-DMTB_JOIN_DIR_QPT=1 ...) get their logic checked here first
        subprocess.check_call(cmd)
    link = os.path.join(out_dir, "libmtb.so")            # tests that compile a program of their own against the library link with -lmtb
    if not os.path.exists(link):
        os.symlink("libmtb_hipemu.so", link)
    exe = os.path.join(out_dir, "mtb_classify")
    drv = os.path.join(ROOT, "metabuli_amd", "csrc", "host", "classify_main.cpp")
    if not os.path.exists(exe) or os.path.getmtime(exe) < max(os.path.getmtime(lib), os.path.getmtime(drv)):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-pthread", "-o", exe, drv, "-L" + out_dir, "-l:libmtb_hipemu.so", "-lz", "-Wl,-rpath," + out_dir] +
                              (["-fsanitize=address"] if asan else []))
    return lib


if __name__ == "__main__":
    print(build(sys.argv[1] if len(sys.argv) > 1 else "/tmp/mtb_hipemu", asan=len(sys.argv) > 2 and sys.argv[2] == "asan"))
