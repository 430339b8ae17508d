import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _hip_device_count():
    """number of HIP devices, probed in a child process (this process must not load a HIP runtime before torch / libmtb
    pick theirs); 0 when there is no runtime / GPU"""
    import subprocess
    code = ("import ctypes\n"
            "n = ctypes.c_int(0)\n"
            "for name in ('libamdhip64.so', 'libamdhip64.so.7', '/opt/rocm/lib/libamdhip64.so'):\n"
            "    try:\n"
            "        lib = ctypes.CDLL(name)\n"
            "    except OSError:\n"
            "        continue\n"
            "    rc = lib.hipGetDeviceCount(ctypes.byref(n))\n"
            "    print(n.value if rc == 0 else 0)\n"
            "    break\n"
            "else:\n"
            "    print(0)\n")
    try:
        out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120).stdout.strip().split("\n")[-1]
        return int(out)
    except Exception:
        return 0


def pytest_collection_modifyitems(config, items):
    """a bare `pytest` on a box without a GPU skips the gpu-marked tests instead of erroring in their fixtures;
    with `-m gpu` on such a box they are still skipped loudly (the driver runs them on an MI355X)."""
    if not any("gpu" in it.keywords for it in items):
        return
    if os.path.exists("/dev/kfd") or _hip_device_count() > 0:       # /dev/kfd: the ROCm compute device node
        return
    if os.environ.get("MTB_HIPEMU") and os.environ.get("MTB_LIB"):   # the library's sources built against tests/hipemu (tests/test_hipemu.py sets this up)
        return
    skip = pytest.mark.skip(reason="no HIP device on this machine (libmtb has no CPU path)")
    for it in items:
        if "gpu" in it.keywords:
            it.add_marker(skip)


@pytest.fixture(scope="session")
def orc():
    from helpers import Oracle
    return Oracle()


@pytest.fixture(scope="session")
def emu():
    from helpers import Emu
    return Emu()


class Toy:
    """A toy database + reads + the oracle's full answer, built once per mode."""

    def __init__(self, orc, tmpdir, syncmer, paired, seed, n_reads=400, length=150, seq_mode=None, err=0.01,
                 lognormal=False, genome_len=30000, with_n=0.1, kmer_format=2, accession_level=0, strain_rank="no rank", genus_div=0.15):
        from helpers import build_toy_db, default_params
        from metabuli_amd import synth
        self.p = default_params(seq_mode=seq_mode or (2 if paired else 1), syncmer=syncmer, kmer_format=kmer_format,
                                accession_level=accession_level)
        self.world = synth.make_world(seed=seed, n_genera=4, species_per_genus=3, strains_per_species=2, genome_len=genome_len,
                                      strain_rank=strain_rank, genus_div=genus_div)
        self.dbdir = str(tmpdir)
        self.values, self.taxids = build_toy_db(orc, self.world, self.p, self.dbdir)
        self.tax = orc.load_taxonomy(os.path.join(self.dbdir, "taxonomy"))
        self.db = orc.open_db(self.dbdir, self.tax, self.p)
        rng = np.random.default_rng(seed + 100)
        out = synth.sample_reads(rng, self.world, n_reads, length=length, err=err, with_n=with_n, paired=paired, lognormal=lognormal)
        if paired:
            self.b1, self.o1, self.b2, self.o2, self.truth = out
        else:
            self.b1, self.o1, self.truth = out
            self.b2 = self.o2 = None
        self.n_reads = n_reads
        self.ref = orc.classify(self.db, self.tax, self.p, self.b1, self.o1, self.b2, self.o2)


class HotToy:
    """A toy database with LONG candidate runs: every fourth metamer of the first genome is also filed under `n_hot` further species of
    its genus, each with the amino-acid part kept and 0-3 codon ids of the DNA part redrawn -- a conserved protein shared by many
    species (runs of 10^3-10^4 candidates in real databases).  A query of that genome then meets a run of > n_hot
    candidates with a spread of hamming sums: the join's wave-cooperative scan (kernels_dir.h, MTB_JOIN_COOP_MIN) instead of its
    per-lane loop."""

    def __init__(self, orc, tmpdir, seq_mode=1, n_reads=150, length=150, n_hot=70, seed=77, err=0.01, lognormal=False, keep=1.0):
        from helpers import build_toy_db, default_params
        from metabuli_amd import synth
        paired = seq_mode == 2
        self.p = default_params(seq_mode=seq_mode, syncmer=1)
        w = synth.make_world(seed=seed, n_genera=2, species_per_genus=2, strains_per_species=1, genome_len=20000, with_euk=False)
        rng = np.random.default_rng(seed)
        nxt = max(w.tax.parent) + 1
        g0 = w.genomes[0][1]
        k, _, _ = orc.extract_batch(default_params(seq_mode=3, syncmer=1), g0, np.array([0, len(g0)], np.uint64))
        v0 = np.unique(k["value"])[::4]
        ev, et = [], []
        for i in range(n_hot):
            w.tax.add(nxt, 4, "species", f"hot{i}")
            dna = v0 & np.uint64(0xFFFFFF)
            for _ in range(3):                         # up to three codon ids redrawn (none for a third of the entries)
                pos = rng.integers(0, 8, size=len(v0)).astype(np.uint64) * np.uint64(3)
                newc = rng.integers(0, 8, size=len(v0)).astype(np.uint64)
                hit = rng.random(len(v0)) < 0.45
                dna = np.where(hit, (dna & ~(np.uint64(7) << pos)) | (newc << pos), dna)
            sel = rng.random(len(v0)) < keep if keep < 1.0 else np.ones(len(v0), bool)     # keep < 1: a hot species files only a sparse subset (most of its matches in a read are alone in their species)
            ev.append(((v0 & ~np.uint64(0xFFFFFF)) | dna)[sel]); et.append(np.full(int(sel.sum()), nxt, np.int32))
            nxt += 1
        self.world = w
        self.dbdir = str(tmpdir); os.makedirs(self.dbdir, exist_ok=True)
        self.values, self.taxids = build_toy_db(orc, w, self.p, self.dbdir, extra=(np.concatenate(ev), np.concatenate(et)))
        aa = self.values >> np.uint64(24)
        self.max_run = int(np.diff(np.flatnonzero(np.concatenate([[True], aa[1:] != aa[:-1], [True]]))).max())
        self.tax = orc.load_taxonomy(os.path.join(self.dbdir, "taxonomy"))
        self.db = orc.open_db(self.dbdir, self.tax, self.p)
        out = synth.sample_reads(np.random.default_rng(seed + 1), w, n_reads, length=length, err=err, frac_random=0.1, paired=paired, lognormal=lognormal)
        if paired:
            self.b1, self.o1, self.b2, self.o2, self.truth = out
        else:
            self.b1, self.o1, self.truth = out
            self.b2 = self.o2 = None
        self.n_reads = n_reads
        self.ref = orc.classify(self.db, self.tax, self.p, self.b1, self.o1, self.b2, self.o2)


TOY_MODES = {
    "sync_se": dict(syncmer=1, paired=False, seed=1),
    "dense_se": dict(syncmer=0, paired=False, seed=2),
    "sync_pe": dict(syncmer=1, paired=True, seed=3),
    "dense_pe": dict(syncmer=0, paired=True, seed=4),
    "old_format_pe": dict(syncmer=0, paired=True, seed=6, kmer_format=1),
    "sync_se_acc2": dict(syncmer=1, paired=False, seed=7, accession_level=2, strain_rank="accession"),
    "sync_long": dict(syncmer=1, paired=False, seed=5, n_reads=40, length=3000, seq_mode=3, err=0.05, lognormal=True),
    # segments beyond one LDS sort chunk (8192 matches): chunk sort + one and two merge passes
    "sync_xlong": dict(syncmer=1, paired=False, seed=8, n_reads=14, length=22000, seq_mode=3, err=0.03, lognormal=True),
}


@pytest.fixture(scope="session", params=list(TOY_MODES))
def toy(request, orc, tmp_path_factory):
    return Toy(orc, tmp_path_factory.mktemp(request.param), **TOY_MODES[request.param])
