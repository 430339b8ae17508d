#!/usr/bin/env python3
"""Why reads leave the register-resident scorer: one bench-like batch with libmtb_dbg.so (make -C metabuli_amd/csrc libmtb_dbg.so).
Usage (GPU box): MTB_LIB=metabuli_amd/csrc/libmtb_dbg.so python profiles/scripts/fast_reasons.py [reads] [targets] [seq_mode 1|2]"""
import ctypes as C, os, sys, tempfile
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import bench, metabuli_amd as M
reads = int(float(sys.argv[1])) if len(sys.argv) > 1 else 1_000_000
targets = int(float(sys.argv[2])) if len(sys.argv) > 2 else int(2e9)
mode = int(sys.argv[3]) if len(sys.argv) > 3 else 1
dev = torch.device("cuda", 0)
ctx = M.Context(0)
params = M.default_params(seq_mode=mode, syncmer=1, smer_len=5)
world = bench.build_world(1234, 24, 1_000_000, 130_000)
taxdir = tempfile.mkdtemp(); world.tax.write(taxdir)
rv, rt, _ = bench.extract_targets(ctx, M, world, params)
T_cap = targets + len(rv)
dv = torch.empty(T_cap, dtype=torch.int64, device=dev); di = torch.empty(T_cap, dtype=torch.int32, device=dev)
T = ctx.synth_index(1234, targets, world.filler_tax_lo, world.filler_tax_hi, rv, rt, dv.data_ptr(), di.data_ptr())
tl = np.concatenate([np.unique(rt), np.arange(world.filler_tax_lo, world.filler_tax_hi + 1, dtype=np.int32)])
ix = ctx.index_from_device(dv.data_ptr(), di.data_ptr(), T, taxdir, tl, params)
b2 = None
if mode == 2:
    b, o, b2 = bench.gen_reads(torch, dev, world.genomes, reads, 150, 0.10, 0.005, 1234 + 17, paired=True)
else:
    b, o = bench.gen_reads(torch, dev, world.genomes, reads, 150, 0.10, 0.005, 1234 + 17)
res = torch.empty(reads * 24, dtype=torch.uint8, device=dev)
cap = reads * 40 * mode + 1024
tt = torch.empty(cap, dtype=torch.int32, device=dev); tc = torch.empty(cap, dtype=torch.int32, device=dev)
out = (C.c_ulonglong * 32)()
for it in range(2):
    ctx.classify_batch_device(ix, params, b.data_ptr(), o.data_ptr(), b2.data_ptr() if b2 is not None else 0, o.data_ptr() if b2 is not None else 0,
                              reads, reads * 150 * mode, res.data_ptr(), tt.data_ptr(), tc.data_ptr(), cap)
    M.lib().mtb_debug_fast_reasons(ctx.h, out)
v = list(out)
names = ["tail overflow / buckets", "> 8 species (pairs: > 24 runs, or > 320 matches)", "S1 not sorted", "S2 group of 2", "S3 > 64 paths", "handled", "sum emitted paths", "sum species with paths"]
for n, x in zip(names, v):
    print(f"{n:28s} {x:12d}  {x / reads:.4f} per read")
print("generic:", ctx.last_stats().n_generic_reads)
ph = ["setup+issue loads", "order+keys", "flags+links", "chain", "emission", "combination", "decision", "filter", "gather", "descent+output"]
tot = sum(v[8:18])
for n, x in zip(ph, v[8:18]):
    print(f"phase {n:22s} {x / reads:10.0f} cycles/read  {100 * x / max(tot, 1):5.1f} %")
