# Mutation fuzzing of the taxonomy loaders (host_db.h: taxonomyDB binary, *.dmp) under AddressSanitizer + UBSan, through tests/emu/taxonomy_check.cpp:
# exit code 0 or 1, no sanitizer report, no hang.  Not collected by pytest:  python tests/fuzz/fuzz_taxonomy.py [seed] [cases]
import random, shutil, struct
import os, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
D = tempfile.mkdtemp(prefix="mtb_fuzz_")
SAN = ["g++", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-std=c++17", "-pthread"]
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, ROOT)
import taxdb_writer as tw
from metabuli_amd import synth
exe = os.path.join(D, "taxonomy_check_asan")
subprocess.check_call(SAN + ["-o", exe, os.path.join(ROOT, "tests", "emu", "taxonomy_check.cpp")])
w = synth.make_world(seed=5, n_genera=4, species_per_genus=3, strains_per_species=2, genome_len=100)
lines = [(t, w.tax.parent[t], w.tax.rank[t]) for t in sorted(w.tax.parent)]; names = {t: w.tax.name[t] for t in w.tax.parent}
good = {}
for ui in (True, False):
    p = os.path.join(D, f"good_{ui}"); tw.write_taxonomy_db(p, lines, names, use_internal=ui); good[ui] = open(p, 'rb').read()
w.tax.write(os.path.join(D, "dmp"))
dmp = {f: open(os.path.join(D, "dmp", f), 'rb').read() for f in ("nodes.dmp", "names.dmp", "merged.dmp")}
rnd = random.Random(int(sys.argv[1]) if len(sys.argv) > 1 else 1); N = int(sys.argv[2]) if len(sys.argv) > 2 else 300
bad = 0
def mutate(b, text=False):
    b = bytearray(b); k = rnd.choice(["flip", "trunc", "zero", "i32", "insert", "many"])
    if not b: return bytes(b), k
    if k == "flip": p = rnd.randrange(len(b)); b[p] ^= 1 << rnd.randrange(8)
    elif k == "many":
        for _ in range(rnd.randrange(2, 30)): p = rnd.randrange(len(b)); b[p] = rnd.randrange(256) if not text else rnd.choice(b"0123456789\t|-\n x")
    elif k == "trunc": b = b[:rnd.randrange(0, len(b))]
    elif k == "zero": p = rnd.randrange(len(b)); n = rnd.randrange(1, 64); b[p:p+n] = bytes(min(n, len(b)-p))
    elif k == "i32":
        if text:
            p = rnd.randrange(len(b)); b[p:p] = rnd.choice([b"-5", b"2147483647", b"99999999999999999999", b"4000000000", b"7"])
        else:
            p = rnd.randrange(0, max(1, len(b) - 4)) & ~3; b[p:p+4] = struct.pack("<i", rnd.choice([-1, 0, 1, 2**31-1, -2**31, 1 << 20, 1 << 29, 7]))
    elif k == "insert": p = rnd.randrange(len(b)); b[p:p] = bytes(rnd.randrange(256) for _ in range(rnd.randrange(1, 40))) if not text else b"5\t|\t6\t|\tspecies\t|\n6\t|\t5\t|\tgenus\t|\n"
    return bytes(b), k
def run(cmd, tag, data):
    global bad
    try: r = subprocess.run(cmd, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, timeout=60)
    except subprocess.TimeoutExpired:
        bad += 1; print("HANG", tag, flush=True); open(os.path.join(D, f"hang_tax_{bad}"), "wb").write(data); return
    if r.returncode not in (0, 1) or b"Sanitizer" in r.stderr or b"runtime error" in r.stderr:
        bad += 1; print("CRASH", tag, r.returncode, r.stderr.decode(errors="replace")[:1200], flush=True); open(os.path.join(D, f"crash_tax_{bad}"), "wb").write(data)
for it in range(N):
    if rnd.random() < 0.5:
        ui = rnd.random() < 0.5; data, k = mutate(good[ui]); p = os.path.join(D, "case_db"); open(p, "wb").write(data)
        run([exe, "db", p], ("db", ui, k), data)
    else:
        d = os.path.join(D, "case_dmp"); shutil.rmtree(d, ignore_errors=True); os.makedirs(d)
        which = rnd.choice(list(dmp)); k = None
        for f, b in dmp.items():
            if f == which: b, k = mutate(b, text=True)
            open(os.path.join(d, f), "wb").write(b)
        run([exe, "dmp", d], ("dmp", which, k), b"")
print("done", N, "cases,", bad, "bad")
