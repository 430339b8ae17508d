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
