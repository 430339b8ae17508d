# The HOST pipeline of the driver alone (parse -> batches -> rows -> file), timed without a GPU: classify_main.cpp linked against
# tests/null_engine/null_mtb.cpp in its MTB_NULL_FAST mode (no look at the bases; optional simulated device time per 1000 reads).
# What it says: the rate the host side can sustain on this machine's cores, i.e. the ceiling the GPU stage is fed at, and what the
# pipeline's fill and drain cost around a GPU stage of a given speed.  Not a classification benchmark: nothing is classified.
# usage: python profiles/scripts/host_pipeline_null_engine.py [n_reads] [threads] [max_reads, comma list] [us of device time per 1000 reads]
import os, shutil, subprocess, sys, tempfile, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
N = int(float(sys.argv[1])) if len(sys.argv) > 1 else 10_000_000
TH = sys.argv[2] if len(sys.argv) > 2 else str(os.cpu_count())
MR = (sys.argv[3] if len(sys.argv) > 3 else "2000000").split(",")
US = sys.argv[4] if len(sys.argv) > 4 else "0"
L = 150
work = tempfile.mkdtemp(prefix="mtb_null_", dir="/dev/shm" if shutil.disk_usage("/dev/shm").free > N * 800 else None)
try:
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-shared", "-fPIC", "-o", os.path.join(work, "libmtb_null.so"), os.path.join(ROOT, "tests/null_engine/null_mtb.cpp"), "-lz"])
    exe = os.path.join(work, "mtb_classify_null")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-pthread", "-o", exe, os.path.join(ROOT, "metabuli_amd/csrc/host/classify_main.cpp"),
                           "-L" + work, "-l:libmtb_null.so", "-lz", "-Wl,-rpath," + work])
    db = os.path.join(work, "db"); os.makedirs(os.path.join(db, "taxonomy"))
    sys.path.insert(0, ROOT)
    from metabuli_amd import synth
    w = synth.make_world(seed=5, n_genera=8, species_per_genus=4, strains_per_species=2, genome_len=100)
    w.tax.write(os.path.join(db, "taxonomy"))
    with open(os.path.join(db, "taxID_list"), "w") as f:
        f.write("".join(f"{t}\n" for t in sorted(t for t in w.tax.parent if t not in set(w.tax.parent.values()))))
    fq = os.path.join(work, "reads.fq")
    rng = np.random.default_rng(1)
    with open(fq, "wb") as f:
        for c0 in range(0, N, 2_000_000):
            n = min(2_000_000, N - c0)
            rec = np.empty((n, 1 + 8 + 1 + L + 3 + L + 1), np.uint8)
            rec[:, 0] = ord("@"); rec[:, 1:9] = np.char.zfill(np.arange(c0, c0 + n).astype("U8"), 8).astype("S8").view(np.uint8).reshape(n, 8)
            rec[:, 9] = 10; rec[:, 10:10 + L] = np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, size=(n, L), dtype=np.uint8)]
            rec[:, 10 + L] = 10; rec[:, 11 + L] = ord("+"); rec[:, 12 + L] = 10; rec[:, 13 + L:13 + 2 * L] = ord("I"); rec[:, 13 + 2 * L] = 10
            rec.tofile(f)
    out = os.path.join(work, "out"); os.makedirs(out)
    env = dict(os.environ, MTB_NULL_FAST="1", MTB_NULL_US_PER_KREAD=US)
    for mr in MR:
        for rep in range(2):
            for fn in os.listdir(out):
                os.remove(os.path.join(out, fn))
            t0 = time.perf_counter()
            r = subprocess.run([exe, "--seq-mode", "1", "--threads", TH, "--max-reads", mr, fq, db, out, "job"], stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, text=True, env=env)
            dt = time.perf_counter() - t0
            print(r.stderr.strip().split("\n")[-1], flush=True)
            if r.returncode:
                raise SystemExit(r.stderr)
            print(f"max-reads {mr}, run {rep}: {N} reads ({os.path.getsize(fq) / 2**20:.0f} MiB FASTQ -> {os.path.getsize(os.path.join(out, 'job_classifications.tsv')) / 2**20:.0f} MiB of rows) "
                  f"in {dt:.2f} s = {N / dt / 1e6:.2f} Mreads/s through the host pipeline alone, {TH} host threads, {US} us of simulated device time per 1000 reads", flush=True)
finally:
    shutil.rmtree(work, ignore_errors=True)
