# A GTDB-scale database FROM FILES, end to end: synthesize T targets on the device, write them in the reference's
# on-disk format with the device-side coder (mtb_index_write) to /dev/shm (the page cache of the box: local disk would be the same bytes
# through the same cache), then run the stand-alone driver (open = chunked decode + pack on load, then classify N reads from a FASTQ file).
# usage: python profiles/scripts/e2e_big.py [targets] [n_reads] [threads] [max_reads per host batch, comma list]
# environment: E2E_WORLD=heavy (2400 genomes, conserved segments, shared-run extras: the bench's heavy-tailed index), E2E_VARIANTS="|--async-results 1" (driver flag sets),
#              E2E_REPS (runs per setting, default 2)
import os, shutil, subprocess, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import bench, metabuli_amd as M
T_WANT = int(float(sys.argv[1])) if len(sys.argv) > 1 else 8_000_000_000
N = int(float(sys.argv[2])) if len(sys.argv) > 2 else 60_000_000
TH = sys.argv[3] if len(sys.argv) > 3 else "64"
MR = (sys.argv[4] if len(sys.argv) > 4 else "2000000,4000000").split(",")
EXTRA = [x.split() for x in os.environ.get("E2E_VARIANTS", "|--async-results 1").split("|")]      # driver flag sets to compare, '|'-separated ("" = defaults)
dev = torch.device("cuda", 0)
ctx = M.Context(0)
params = M.default_params(seq_mode=1, syncmer=1, smer_len=5)
# the box's local disk if it has room for the database + the reads + the output (its page cache then holds them: 3 TB of RAM), else /dev/shm
need = T_WANT * 9.6 + N * 330 + N * 70
base = "/tmp" if shutil.disk_usage("/tmp").free > 1.3 * need else "/dev/shm"
work = os.path.join(base, "mtb_e2e_big")
print(f"working directory {work} ({shutil.disk_usage(base).free / 2**30:.0f} GiB free; {need / 2**30:.0f} GiB needed)", flush=True)
shutil.rmtree(work, ignore_errors=True)
db = os.path.join(work, "db"); os.makedirs(os.path.join(db, "taxonomy"))
HEAVY = os.environ.get("E2E_WORLD", "") == "heavy"      # the bench's default world: 2400 genomes with conserved segments + shared-run extras (heavy-tailed candidate runs)
if HEAVY:
    world = bench.build_world_fast(torch, dev, 1234, 2400, 1_000_000, 130_000, conserved=True)
    world.tax.write(os.path.join(db, "taxonomy"))
    rv, rt, n_extras = bench.extract_targets(ctx, M, world, params, torch, dev, hot_min=8, seed=1234)
    print(f"heavy-tailed world: {len(world.genomes)} genomes, {len(rv) - n_extras} genome-derived targets + {n_extras} shared-run extras", flush=True)
else:
    world = bench.build_world(1234, 8, 500000, 5000)
    world.tax.write(os.path.join(db, "taxonomy"))
    rv, rt, _ = bench.extract_targets(ctx, M, world, params)
NF = T_WANT - len(rv)
dv = torch.empty(T_WANT, dtype=torch.int64, device=dev); di = torch.empty(T_WANT, dtype=torch.int32, device=dev)
t0 = time.perf_counter()
T = ctx.synth_index(1234, NF, world.filler_tax_lo, world.filler_tax_hi, rv, rt, dv.data_ptr(), di.data_ptr())
tl = np.concatenate([np.unique(rt), np.arange(world.filler_tax_lo, world.filler_tax_hi + 1, dtype=np.int32)])
ix = ctx.index_from_device(dv.data_ptr(), di.data_ptr(), T, os.path.join(db, "taxonomy"), tl, params)
print(f"synthetic index: {T} targets in {time.perf_counter() - t0:.1f} s", flush=True)
t0 = time.perf_counter(); ix.write(db); t_write = time.perf_counter() - t0
sz = {f: os.path.getsize(os.path.join(db, f)) for f in ("diffIdx", "info", "split")}
print(f"database: {T} metamers written in {t_write:.1f} s = {(sz['diffIdx'] + sz['info']) / t_write / 1e9:.2f} GB/s ({sz['diffIdx'] / 2**30:.2f} GiB diffIdx, {sz['info'] / 2**30:.2f} GiB info) to {db}", flush=True)
L = 150
fq = os.path.join(work, "reads.fq")
with open(fq, "wb") as f:
    for c0 in range(0, N, 10_000_000):
        n = min(10_000_000, N - c0)
        bases, _ = bench.gen_reads(torch, dev, world.genomes, n, L, 0.10, 0.005, 99 + c0)
        b = bases.cpu().numpy().reshape(n, L)
        name = np.char.zfill(np.arange(c0, c0 + n).astype("U8"), 8).astype("S8").view(np.uint8).reshape(n, 8)
        rec = np.empty((n, 1 + 8 + 1 + L + 3 + L + 1), np.uint8)
        rec[:, 0] = ord("@"); rec[:, 1:9] = name; rec[:, 9] = 10; rec[:, 10:10 + L] = b
        rec[:, 10 + L] = 10; rec[:, 11 + L] = ord("+"); rec[:, 12 + L] = 10; rec[:, 13 + L:13 + 2 * L] = ord("I"); rec[:, 13 + 2 * L] = 10
        rec.tofile(f)
        del bases, b, rec
ix.close(); del ix, dv, di; ctx.close(); torch.cuda.empty_cache()
out = os.path.join(work, "out"); os.makedirs(out)
exe = os.path.join(os.path.dirname(M.LIB_PATH), "mtb_classify")
ref_sum = None
for mr in MR:
  for extra in EXTRA:
    for rep in range(int(os.environ.get("E2E_REPS", "2"))):
        for fn in os.listdir(out):              # every run writes into an empty directory (truncating the previous run's GBs of rows costs tenths of a second)
            os.remove(os.path.join(out, fn))
        t0 = time.perf_counter()
        r = subprocess.run([exe, "--seq-mode", "1", "--threads", TH, "--max-reads", mr] + extra + [fq, db, out, "job"], stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, text=True)
        dt = time.perf_counter() - t0
        print(r.stderr.strip(), flush=True)
        if r.returncode:
            raise SystemExit(f"mtb_classify failed ({r.returncode})")
        import hashlib
        h = hashlib.md5()
        with open(os.path.join(out, "job_classifications.tsv"), "rb") as f:
            for blk in iter(lambda: f.read(1 << 24), b""):
                h.update(blk)
        ref_sum = ref_sum or h.hexdigest()
        print(f"max-reads {mr} {' '.join(extra) or '(defaults)'}, run {rep}: {N} reads ({os.path.getsize(fq) / 2**20:.0f} MiB FASTQ) end to end in {dt:.2f} s = {N / dt / 1e6:.2f} Mreads/s "
              f"(includes opening the database of {T} targets from {(sz['diffIdx'] + sz['info']) / 2**30:.1f} GiB of files), {TH} host threads; rows md5 {h.hexdigest()[:12]}"
              f"{'' if h.hexdigest() == ref_sum else ' DIFFERS FROM THE FIRST RUN'}", flush=True)
print(open(os.path.join(out, "job_report.tsv")).read()[:400])
shutil.rmtree(work, ignore_errors=True)
