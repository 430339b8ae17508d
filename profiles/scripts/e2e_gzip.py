# end-to-end timing of the stand-alone driver on an ORDINARY gzip file (one member, one deflate stream -- written pigz-style: pieces
# compressed independently and joined with sync flushes) against the same reads as plain text
# usage: python profiles/scripts/e2e_gzip.py [n_reads] [n_filler] [threads]
import os, sys, time, subprocess, tempfile, zlib, struct
from concurrent.futures import ThreadPoolExecutor
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import bench, metabuli_amd as M
N = int(float(sys.argv[1])) if len(sys.argv) > 1 else 2_000_000
NF = int(float(sys.argv[2])) if len(sys.argv) > 2 else 200_000_000
TH = sys.argv[3] if len(sys.argv) > 3 else "8"
dev = torch.device("cuda", 0)
ctx = M.Context(0)
params = M.default_params(seq_mode=1, syncmer=1, smer_len=5)
work = tempfile.mkdtemp(prefix="mtb_e2egz_")
db = os.path.join(work, "db"); os.makedirs(os.path.join(db, "taxonomy"))
world = bench.build_world(1234, 8, 500000, 5000)
world.tax.write(os.path.join(db, "taxonomy"))
rv, rt, _ = bench.extract_targets(ctx, M, world, params)
Tc = NF + len(rv)
dv = torch.empty(Tc, dtype=torch.int64, device=dev); di = torch.empty(Tc, dtype=torch.int32, device=dev)
T = ctx.synth_index(1234, NF, world.filler_tax_lo, world.filler_tax_hi, rv, rt, dv.data_ptr(), di.data_ptr())
tl = np.concatenate([np.unique(rt), np.arange(world.filler_tax_lo, world.filler_tax_hi + 1, dtype=np.int32)])
ix = ctx.index_from_device(dv.data_ptr(), di.data_ptr(), T, os.path.join(db, "taxonomy"), tl, params)
ix.write(db)
L = 150
bases, _ = bench.gen_reads(torch, dev, world.genomes, N, L, 0.10, 0.005, 99)
b = bases.cpu().numpy().reshape(N, L)
name = np.char.zfill(np.arange(N).astype("U8"), 8).astype("S8").view(np.uint8).reshape(N, 8)
rec = np.empty((N, 1 + 8 + 1 + L + 3 + L + 1), np.uint8)
rec[:, 0] = ord("@"); rec[:, 1:9] = name; rec[:, 9] = 10; rec[:, 10:10 + L] = b
rec[:, 10 + L] = 10; rec[:, 11 + L] = ord("+"); rec[:, 12 + L] = 10
rng = np.random.default_rng(3)
qblock = rng.choice(np.frombuffer(b"FFFFFFFF:::,,#", np.uint8), size=(100000, L))     # qualities with some entropy (a 100 k-read block, repeated)
for i in range(0, N, 100000): rec[i:i + 100000, 13 + L:13 + 2 * L] = qblock[:min(100000, N - i)]
rec[:, 13 + 2 * L] = 10
fq = os.path.join(work, "reads.fq"); rec.tofile(fq)
del ix, dv, di, bases; ctx.close(); torch.cuda.empty_cache()
raw = rec.reshape(-1)
PIECE = 64 << 20
pieces = [raw[i:i + PIECE] for i in range(0, len(raw), PIECE)]
def comp(k):
    co = zlib.compressobj(4, zlib.DEFLATED, -15)
    data = co.compress(pieces[k])
    return data + (co.flush(zlib.Z_FINISH) if k == len(pieces) - 1 else co.flush(zlib.Z_SYNC_FLUSH))
t0 = time.perf_counter()
with ThreadPoolExecutor(64) as ex: parts = list(ex.map(comp, range(len(pieces))))
crc = 0
for pc in pieces: crc = zlib.crc32(pc, crc)
gz = fq + ".gz"
with open(gz, "wb") as f:
    f.write(b"\x1f\x8b\x08\x00\x00\x00\x00\x00\x00\x03")
    for d in parts: f.write(d)
    f.write(struct.pack("<II", crc & 0xFFFFFFFF, len(raw) & 0xFFFFFFFF))
print(f"{N} reads: {os.path.getsize(fq) / 2**20:.0f} MiB of FASTQ, {os.path.getsize(gz) / 2**20:.0f} MiB as one gzip stream (written in {time.perf_counter() - t0:.1f} s)", flush=True)
out = os.path.join(work, "out"); os.makedirs(out)
exe = os.path.join(os.path.dirname(M.LIB_PATH), "mtb_classify")
for label, tag, path, env in (("plain text", "p", fq, None), ("gzip, block-parallel inflate (pgzip.h)", "g", gz, None), ("gzip, one zlib stream (MTB_NO_PGZIP)", "z", gz, dict(os.environ, MTB_NO_PGZIP="1"))):
    t0 = time.perf_counter()
    subprocess.check_call([exe, "--seq-mode", "1", "--threads", TH, "--max-reads", "2000000", path, db, out, "job_" + tag], stdout=subprocess.DEVNULL, env=env)
    print(f"{label}: end to end {time.perf_counter() - t0:.2f} s", flush=True)
a = open(os.path.join(out, "job_p_classifications.tsv"), "rb").read()
for tag in ("g", "z"):
    assert open(os.path.join(out, f"job_{tag}_classifications.tsv"), "rb").read() == a, tag
print("classifications identical for all three inputs")
