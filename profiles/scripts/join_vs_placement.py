#!/usr/bin/env python3
"""Does the join's spread follow the placement of the slot buffer?  One process, the bench workload; between rounds the slot buffer
is released and re-allocated somewhere else (libmtb_place.so: make -C metabuli_amd/csrc libmtb_xplace.so X=-DMTB_PLACEMENT_DEBUG).
Usage (GPU box): MTB_LIB=metabuli_amd/csrc/libmtb_xplace.so python profiles/scripts/join_vs_placement.py [rounds] [targets]"""
import ctypes as C, os, sys, tempfile
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import bench, metabuli_amd as M
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 6
targets = int(float(sys.argv[2])) if len(sys.argv) > 2 else int(16e9)
reads = 10_000_000
dev = torch.device("cuda", 0)
ctx = M.Context(0)
params = M.default_params(seq_mode=1, syncmer=1, smer_len=5)
world = bench.build_world(1234, 24, 1_000_000, 130_000)
taxdir = tempfile.mkdtemp(); world.tax.write(taxdir)
rv, rt, _ = bench.extract_targets(ctx, M, world, params)
T_cap = targets + len(rv)
dv = torch.empty(T_cap, dtype=torch.int64, device=dev); di = torch.empty(T_cap, dtype=torch.int32, device=dev)
T = ctx.synth_index(1234, targets, world.filler_tax_lo, world.filler_tax_hi, rv, rt, dv.data_ptr(), di.data_ptr())
tl = np.concatenate([np.unique(rt), np.arange(world.filler_tax_lo, world.filler_tax_hi + 1, dtype=np.int32)])
ix = ctx.index_from_device(dv.data_ptr(), di.data_ptr(), T, taxdir, tl, params)
ix.seal(); del di; torch.cuda.empty_cache()
b, o = bench.gen_reads(torch, dev, world.genomes, reads, 150, 0.10, 0.005, 1234 + 17)
res = torch.empty(reads * 24, dtype=torch.uint8, device=dev)
cap = reads * 40 + 1024
tt = torch.empty(cap, dtype=torch.int32, device=dev); tc = torch.empty(cap, dtype=torch.int32, device=dev)
L = M.lib()
L.mtb_debug_move_buffer.argtypes = [C.c_void_p, C.c_char_p, C.c_ulonglong]
def run(n):
    js = []
    for _ in range(n):
        ctx.classify_batch_device(ix, params, b.data_ptr(), o.data_ptr(), 0, 0, reads, reads * 150, res.data_ptr(), tt.data_ptr(), tc.data_ptr(), cap)
        js.append(ctx.last_stats().ms_join)
    return js
print("initial placement: join ms", ["%.1f" % x for x in run(4)[1:]])
for r in range(rounds):
    which = b"segm" if r % 2 == 0 else b"kmersB"
    L.mtb_debug_move_buffer(ctx.h, which, (1 + r) << 30)
    print(f"round {r}: moved {which.decode()} (pad {(1 + r)} GiB): join ms", ["%.1f" % x for x in run(3)[1:]])
