# Mutation fuzzing of the host readers of untrusted input (pgzip.h, fastx.h) under AddressSanitizer + UBSan: every case must end with exit
# code 0 or 1 (an error message), never a sanitizer report, another exit code or a hang.  Not collected by pytest (minutes of CPU):
#   python tests/fuzz/fuzz_readers.py [seed] [cases]
import gzip, zlib, random, struct
import numpy as np
import os, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
D = tempfile.mkdtemp(prefix="mtb_fuzz_")
SAN = ["g++", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-std=c++17", "-pthread"]
for src, exe in (("pgzip_check.cpp", "pgzip_check_asan"), ("fastx_dump.cpp", "fastx_dump_asan")):
    subprocess.check_call(SAN + ["-o", os.path.join(D, exe), os.path.join(ROOT, "tests", "emu", src), "-lz"])
def fastq(n, seed, L=150):
    r = np.random.default_rng(seed)
    bases = r.choice(np.frombuffer(b"ACGT", np.uint8), size=(n, L)); qual = r.choice(np.frombuffer(b"FFFFFFFF:::,,#", np.uint8), size=(n, L))
    return b"".join(b"@read%d len=%d\n" % (i, L) + bases[i].tobytes() + b"\n+\n" + qual[i].tobytes() + b"\n" for i in range(n))
def bgzf(data, blk=60000):
    out=[]
    for a in range(0, len(data), blk):
        c = zlib.compressobj(6, zlib.DEFLATED, -15); d = c.compress(data[a:a+blk]) + c.flush()
        bsize = len(d) + 25
        out.append(b"\x1f\x8b\x08\x04\0\0\0\0\0\xff\x06\0BC\x02\0" + struct.pack("<H", bsize) + d + struct.pack("<II", zlib.crc32(data[a:a+blk]), len(data[a:a+blk])))
    out.append(bytes.fromhex("1f8b08040000000000ff0600424302001b0003000000000000000000"))
    return b"".join(out)
text = fastq(4000, 1)
cases = {"gz6": gzip.compress(text, 6), "gz1": gzip.compress(text, 1), "bgzf": bgzf(text), "multi": gzip.compress(text[:300000], 9) + gzip.compress(text[300000:], 1), "plain": text}
rnd = random.Random(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
N = int(sys.argv[2]) if len(sys.argv) > 2 else 200
bad = 0
def mutate(b):
    b = bytearray(b)
    k = rnd.choice(["flip", "trunc", "zero", "dup", "insert", "flipmany", "header"])
    if k == "flip":
        p = rnd.randrange(len(b)); b[p] ^= 1 << rnd.randrange(8)
    elif k == "flipmany":
        for _ in range(rnd.randrange(2, 50)):
            p = rnd.randrange(len(b)); b[p] ^= 1 << rnd.randrange(8)
    elif k == "trunc":
        b = b[:rnd.randrange(1, len(b))]
    elif k == "zero":
        p = rnd.randrange(len(b)); n = rnd.randrange(1, 5000); b[p:p+n] = bytes(min(n, len(b)-p))
    elif k == "dup":
        p = rnd.randrange(len(b)); n = rnd.randrange(1, 70000); b[p:p] = b[p:p+n]
    elif k == "insert":
        p = rnd.randrange(len(b)); b[p:p] = bytes(rnd.randrange(256) for _ in range(rnd.randrange(1, 300)))
    elif k == "header":
        p = rnd.randrange(min(40, len(b))); b[p] = rnd.randrange(256)
    return bytes(b), k
for it in range(N):
    name = rnd.choice(list(cases))
    data, kind = mutate(cases[name])
    ext = ".fq" if name == "plain" else ".fq.gz"
    path = os.path.join(D, "case" + ext)
    open(path, "wb").write(data)
    cmds = []
    if name != "plain" and name != "bgzf":
        cmds.append([os.path.join(D, "pgzip_check_asan"), path, str(rnd.choice([1, 3, 8])), str(rnd.choice([65536, 70001, 200000]))])
    cmds.append([os.path.join(D, "fastx_dump_asan"), path, str(rnd.choice([1, 4, 8])), str(rnd.choice([65536, 300000, 1 << 22])), str(rnd.choice([100, 1000, 100000]))] + (["pack"] if rnd.random() < 0.5 else []))
    for cmd in cmds:
        try:
            r = subprocess.run(cmd, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, timeout=120)
        except subprocess.TimeoutExpired:
            bad += 1; keep = os.path.join(D, f"hang_{it}{ext}"); open(keep, "wb").write(data); print("HANG", name, kind, cmd, keep, flush=True); continue
        if r.returncode not in (0, 1) or b"Sanitizer" in r.stderr or b"runtime error" in r.stderr:
            bad += 1; keep = os.path.join(D, f"crash_{it}{ext}"); open(keep, "wb").write(data)
            print("CRASH", name, kind, cmd, r.returncode, keep, r.stderr.decode(errors="replace")[:1500], flush=True)
print("done", N, "cases,", bad, "bad")
