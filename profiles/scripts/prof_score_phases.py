# profiling-build driver: k_score phase cycles on the small bench workload
import os, sys, ctypes as C, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import bench, metabuli_amd as M
dev = torch.device('cuda', 0)
ctx = M.Context(0)
SEQ_MODE = int(os.environ.get('PROF_SEQ_MODE', '1')); RLEN = int(os.environ.get('PROF_READ_LEN', '150')); NREADS = int(os.environ.get('PROF_READS', '500000'))
params = M.default_params(seq_mode=SEQ_MODE, syncmer=1, smer_len=5)
import tempfile
world = bench.build_world(1234, 8, 500000, 5000)
taxdir = tempfile.mkdtemp(); world.tax.write(taxdir)
rv, rt, _ = bench.extract_targets(ctx, M, world, params)
nf = int(2e8); Tc = nf + len(rv)
dv = torch.empty(Tc, dtype=torch.int64, device=dev); di = torch.empty(Tc, dtype=torch.int32, device=dev)
T = ctx.synth_index(1234, nf, world.filler_tax_lo, world.filler_tax_hi, rv, rt, dv.data_ptr(), di.data_ptr())
tl = np.concatenate([np.unique(rt), np.arange(world.filler_tax_lo, world.filler_tax_hi + 1, dtype=np.int32)])
ix = ctx.index_from_device(dv.data_ptr(), di.data_ptr(), T, taxdir, tl, params)
N = NREADS
db, do = bench.gen_reads(torch, dev, world.genomes, N, RLEN, 0.10, 0.005, 99)
dres = torch.empty(N * 24, dtype=torch.uint8, device=dev); cap = N * (20 + RLEN // 9) + 1024
dtt = torch.empty(cap, dtype=torch.int32, device=dev); dtc = torch.empty(cap, dtype=torch.int32, device=dev)
out = (C.c_ulonglong * 24)()
for it in range(2):
    ctx.classify_batch_device(ix, params, db.data_ptr(), do.data_ptr(), 0, 0, N, N * RLEN, dres.data_ptr(), dtt.data_ptr(), dtc.data_ptr(), cap)
    M.lib().mtb_debug_phase_cycles(ctx.h, out)
st = ctx.last_stats()
tot = sum(out[:16])
names = ["load+keys", "rank", "permute", "flags+ids", "starts", "links", "chainDP", "emit", "combine", "select", "filter", "gather", "climb", "decide+out", "slot load+compact", "read setup"]
print("score ms", st.ms_score, "total cycles/read", int(tot / N))
for k, nm in enumerate(names):
    print(f"  {nm:12s} {int(out[k] / N):7d} cycles/read  {100.0 * out[k] / tot:5.1f} %")

jn = ["load+window", "find (extra pass)", "find+count", "atomic probe (extra)", "reserve+emit"]
nblk = (st.n_kmers + 511) // 512
jt = sum(out[16:21])
print("join ms", st.ms_join, "cycles per 512-query workgroup (thread 0):")
for k, nm in enumerate(jn):
    print(f"  {nm:22s} {int(out[16 + k] / nblk):7d}  {100.0 * out[16 + k] / jt:5.1f} %")
