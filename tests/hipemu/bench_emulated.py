"""TEST INFRASTRUCTURE ONLY: bench.py's own logic (rank handling, the JSON line, the parity legs) executed WITHOUT a GPU, against the library
build that runs on the CPU stand-in of the HIP runtime (tests/hipemu/hip/hip_runtime.h; MTB_LIB points at it).  "Device" memory of that
build is host memory, so torch CPU tensors play the device tensors' part and the torch.cuda calls bench.py makes are no-ops here.
Small sizes, no timing claims.  The product benchmark (bench.py) knows nothing of this: it only lets main() be handed a device.

    MTB_HIPEMU=1 MTB_LIB=/tmp/mtb_hipemu/libmtb_hipemu.so python tests/hipemu/bench_emulated.py --reads 3000 --targets 1.5e6 --species 8 ...
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

if __name__ == "__main__":
    if not (os.environ.get("MTB_HIPEMU") and os.environ.get("MTB_LIB")):
        raise SystemExit("bench_emulated.py needs MTB_HIPEMU=1 and MTB_LIB=<the emulated library> (tests/hipemu/build_emulated.py)")
    import torch
    torch.cuda.synchronize = lambda *a, **k: None
    torch.cuda.set_device = lambda *a, **k: None
    torch.cuda.empty_cache = lambda: None
    torch.cuda.mem_get_info = lambda *a, **k: (64 << 30, 64 << 30)
    import bench
    bench.main(device=torch.device("cpu"))
