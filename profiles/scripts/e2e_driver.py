# end-to-end timing of the stand-alone driver (FASTQ on disk -> TSVs on disk) on a synthetic database written by
# mtb_index_write; usage: python profiles/scripts/e2e_driver.py [n_reads] [n_filler] [threads] [max_reads per host batch, comma list] [gpu workers, comma list]
import os, sys, time, subprocess, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import bench, metabuli_amd as M
N = int(float(sys.argv[1])) if len(sys.argv) > 1 else 2_000_000
NF = int(float(sys.argv[2])) if len(sys.argv) > 2 else 200_000_000
TH = sys.argv[3] if len(sys.argv) > 3 else "8"
MR = (sys.argv[4] if len(sys.argv) > 4 else "2000000").split(",")
GW = (sys.argv[5] if len(sys.argv) > 5 else "2").split(",")
dev = torch.device("cuda", 0)
ctx = M.Context(0)
params = M.default_params(seq_mode=1, syncmer=1, smer_len=5)
work = tempfile.mkdtemp(prefix="mtb_e2e_")
db = os.path.join(work, "db"); os.makedirs(os.path.join(db, "taxonomy"))
world = bench.build_world(1234, 8, 500000, 5000)
world.tax.write(os.path.join(db, "taxonomy"))
rv, rt = bench.extract_targets(ctx, M, world, params)
Tc = NF + len(rv)
dv = torch.empty(Tc, dtype=torch.int64, device=dev); di = torch.empty(Tc, dtype=torch.int32, device=dev)
T = ctx.synth_index(1234, NF, world.filler_tax_lo, world.filler_tax_hi, rv, rt, dv.data_ptr(), di.data_ptr())
tl = np.concatenate([np.unique(rt), np.arange(world.filler_tax_lo, world.filler_tax_hi + 1, dtype=np.int32)])
ix = ctx.index_from_device(dv.data_ptr(), di.data_ptr(), T, os.path.join(db, "taxonomy"), tl, params)
t0 = time.perf_counter(); ix.write(db); t_write = time.perf_counter() - t0
print(f"database: {T} metamers written in {t_write:.1f} s ({os.path.getsize(os.path.join(db, 'diffIdx')) / 2**30:.2f} GiB diffIdx)")
L = 150
bases, _ = bench.gen_reads(torch, dev, world, N, L, 0.10, 0.005, 99)
b = bases.cpu().numpy().reshape(N, L)
name = np.char.zfill(np.arange(N).astype("U8"), 8).astype("S8").view(np.uint8).reshape(N, 8)
rec = np.empty((N, 1 + 8 + 1 + L + 3 + L + 1), np.uint8)
rec[:, 0] = ord("@"); rec[:, 1:9] = name; rec[:, 9] = 10; rec[:, 10:10 + L] = b
rec[:, 10 + L] = 10; rec[:, 11 + L] = ord("+"); rec[:, 12 + L] = 10; rec[:, 13 + L:13 + 2 * L] = ord("I"); rec[:, 13 + 2 * L] = 10
fq = os.path.join(work, "reads.fq"); rec.tofile(fq)
del ix, dv, di, bases; ctx.close(); torch.cuda.empty_cache()
out = os.path.join(work, "out"); os.makedirs(out)
exe = os.path.join(os.path.dirname(M.LIB_PATH), "mtb_classify")
for mr in MR:
 for gw in GW:
  for rep in range(2):
    t0 = time.perf_counter()
    subprocess.check_call([exe, "--seq-mode", "1", "--threads", TH, "--max-reads", mr, "--gpu-workers", gw, fq, db, out, "job"], stdout=subprocess.DEVNULL)
    dt = time.perf_counter() - t0
    print(f"max-reads {mr}, gpu-workers {gw}, run {rep}: {N} reads ({os.path.getsize(fq) / 2**20:.0f} MiB FASTQ) end to end in {dt:.2f} s = {N / dt / 1e6:.2f} Mreads/s "
          f"(includes opening the index: decode {T} metamers, taxonomy), {TH} host threads", flush=True)
print(open(os.path.join(out, "job_report.tsv")).read()[:400])
