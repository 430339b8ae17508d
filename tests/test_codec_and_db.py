"""diffIdx codec and on-disk database format (IndexCreator.cpp:817-892,
KmerMatcher.h:282-297): round trips, group sizes, split checkpoints."""
import os

import numpy as np

from helpers import default_params


def test_diffidx_roundtrip_and_group_sizes(orc):
    rng = np.random.default_rng(0)
    for hi in (1 << 14, 1 << 20, 1 << 40, (1 << 64) - 1):
        v = np.unique(rng.integers(0, hi, size=5000, dtype=np.uint64))
        enc = orc.diffidx_encode(v)
        assert (orc.diffidx_decode(enc) == v).all()
        assert int((enc >> 15).sum()) == len(v)          # one terminator per metamer
    # 1..5 fragments: deltas of 0, 2^15-1, 2^15, 2^30, 2^45, 2^60
    v = np.cumsum(np.array([0, (1 << 15) - 1, 1 << 15, 1 << 30, 1 << 45, 1 << 60], dtype=np.uint64)).astype(np.uint64)
    enc = orc.diffidx_encode(v)
    ends = np.flatnonzero(enc >> 15)
    assert list(np.diff(np.concatenate([[-1], ends]))) == [1, 1, 2, 3, 4, 5]
    assert (orc.diffidx_decode(enc) == v).all()


def test_db_files_and_split_checkpoints(toy, orc):
    d = toy.dbdir
    diff = np.fromfile(os.path.join(d, "diffIdx"), dtype=np.uint16)
    info = np.fromfile(os.path.join(d, "info"), dtype=np.int32)
    split = np.fromfile(os.path.join(d, "split"), dtype=np.uint64).reshape(-1, 3)
    assert len(split) == 4096 and (split[0] == 0).all()
    assert (info == toy.taxids).all() and (orc.diffidx_decode(diff) == toy.values).all()
    ends = np.flatnonzero(diff >> 15)                     # fragment index of every metamer's terminator
    used = split[(split[:, 0] != 0)]
    assert len(used) > 100
    aam = ~np.uint64(0xFFFFFF)
    for ad, doff, ioff in used[:: max(1, len(used) // 50)]:
        j = int(ioff) - 1                                 # entry whose value is ADkmer (KmerMatcher.cpp:256-271)
        assert toy.values[j] == ad and ends[j] + 1 == doff
        assert (toy.values[j] & aam) != (toy.values[j - 1] & aam)   # checkpoints sit on a new amino-acid part
    txt = open(os.path.join(d, "db.parameters")).read()
    assert "Skip_redundancy\t1" in txt and f"Kmer_format\t{toy.p.kmer_format}" in txt


def test_db_parameters_override(toy, emu):
    import ctypes as C
    from helpers import Params
    p = Params(seq_mode=2, syncmer=0, smer_len=5, kmer_format=1, skip_redundancy=0)
    assert emu.lib.emu_load_db_parameters(toy.dbdir.encode(), C.byref(p)) == 0
    assert p.kmer_format == toy.p.kmer_format and p.skip_redundancy == 1 and p.syncmer == toy.p.syncmer
    assert p.smer_len == 5     # the DB writes "Syncmer_len", the loader reads "S-mer_len" 
