"""bench.py's workload generators and parity plumbing on CPU tensors (no GPU, no library): the heavy-tailed candidate runs
(conserved segments, shared-run extras) and the sub-database positions a parity sample needs."""
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def test_conserved_segments_are_shared_at_the_protein_level(orc):
    """the conserved segments give LONG runs of equal amino-acid parts whose DNA differs from genome to genome; without them a run
    holds the species of one genus at most"""
    from helpers import default_params
    dev = torch.device("cpu")
    p = default_params(seq_mode=3, syncmer=1)

    def runs(w):
        vals = []
        for _, g in w.genomes:
            k, _, _ = orc.extract_batch(p, g, np.array([0, len(g)], np.uint64))
            vals.append(np.unique(k["value"]))
        v = np.sort(np.concatenate(vals))
        aa = v >> np.uint64(24)
        edges = np.flatnonzero(np.concatenate([[True], aa[1:] != aa[:-1], [True]]))
        return v, np.diff(edges), edges
    w = bench.build_world_fast(torch, dev, 5, 40, 600_000, 100, conserved=True)
    assert len(w.genomes) == 40 and all(len(g) == 600_000 for _, g in w.genomes)
    assert set(np.unique(np.concatenate([g for _, g in w.genomes[:3]]))) == set(b"ACGT")
    v, rl, edges = runs(w)
    assert rl.max() >= 25 and (rl >= 12).sum() > 2000            # prevalence-1 segments: 78 % of 40 copies keep an 8-mer of the coding frame
    i = int(np.argmax(rl))
    assert len(np.unique(v[edges[i]: edges[i + 1]])) >= rl[i] // 3        # ... with many different DNA parts
    w0 = bench.build_world_fast(torch, dev, 5, 40, 600_000, 100, conserved=False)
    assert runs(w0)[1].max() <= 6


def test_shared_run_extras_keep_amino_acid_part_codons_and_order():
    dev = torch.device("cpu")
    rng = np.random.default_rng(3)
    # genome-derived values of one sign half: 600 amino-acid parts of 8 letters < 21, run lengths 1..40; every letter has 2-4 valid codon ids
    valid = {L: sorted(rng.choice(8, size=int(rng.integers(2, 5)), replace=False).tolist()) for L in range(21)}
    valid[7] = [3]                                                   # a letter with a single codon (Met, Trp): cannot change
    parts = []
    for _ in range(600):
        letters = rng.integers(0, 21, size=8)
        letters[0] = letters[0] % 16                                 # (top bit clear: the non-negative half)
        a = 0
        for L in letters:
            a = (a << 5) | int(L)
        r = int(rng.integers(1, 41))
        dn = set()
        for _ in range(r):
            d = 0
            for L in letters:
                d = (d << 3) | int(rng.choice(valid[int(L)]))
            dn.add(d)
        parts.append((a, letters, sorted(dn)))
    parts.sort(key=lambda x: x[0])
    vals = np.array([(a << 24) | d for a, _, dn in parts for d in dn], dtype=np.int64)
    assert (np.diff(vals) > 0).all()
    hv = torch.from_numpy(vals)
    ev, et = bench.hot_run_extras(torch, dev, hv, 1000, 500, 8, 42)
    ev, et = ev.numpy(), et.numpy()
    assert len(ev) and (np.diff(ev) >= 0).all()
    same = np.diff(ev) == 0
    assert (np.diff(et)[same] >= 0).all()                                        # (value, taxid) order
    assert ((et >= 1000) & (et < 1500)).all()
    e_aa = ev >> 24
    n_hot = 0
    for a, letters, dn in parts:
        mine = ev[e_aa == a]
        if len(dn) < 8:
            assert len(mine) == 0
            continue
        assert len(mine) in (0, len(dn), 3 * len(dn), 7 * len(dn))
        n_hot += len(mine) > 0
        for j, L in enumerate(letters):                                          # every codon id is one the letter has
            got = set(((mine >> (3 * (7 - j))) & 7).tolist())
            assert got <= set(valid[int(L)]), (got, valid[int(L)])
        if len(mine) and not (letters == 7).all():
            assert len(set(mine.tolist()) - {(a << 24) | d for d in dn}) > 0     # ... and the DNA parts are new ones
    assert n_hot > 5
    # negative half (top bit set): same arithmetic
    hv2 = torch.from_numpy(np.sort((vals | np.int64(-2**63))))
    ev2, _ = bench.hot_run_extras(torch, dev, hv2, 1000, 500, 8, 42)
    assert len(ev2) > 0 and bool((ev2 < 0).all()) and bool((ev2[1:] >= ev2[:-1]).all())       # (the multiplier is a hash of the amino-acid part: other counts)


@pytest.mark.parametrize("stride", [0, 7])
def test_closure_positions_hold_every_candidate_and_the_last_entry(stride):
    rng = np.random.default_rng(11)
    T = 50_000
    # unsigned-sorted values with both sign halves, runs of equal amino-acid parts
    aa = np.sort(rng.choice(np.arange(1 << 20, dtype=np.uint64) << np.uint64(20), size=9000, replace=False))       # 40-bit amino-acid parts, some with the top bit
    aa[-2000:] |= np.uint64(1 << 39)
    aa = np.sort(aa)
    v = np.sort((rng.choice(aa, size=T) << np.uint64(24)) | rng.integers(0, 1 << 24, size=T).astype(np.uint64))
    d_values = torch.from_numpy(v.view(np.int64).copy())
    q = np.concatenate([rng.choice(v, size=500), (rng.choice(aa, size=50) << np.uint64(24)) | np.uint64(5), rng.integers(0, 2**63, size=100).astype(np.uint64)])
    pos = bench.closure_positions(torch, d_values, T, q, stride).numpy()
    assert (np.diff(pos) > 0).all() and pos[-1] == T - 1
    want = np.isin(v >> np.uint64(24), np.unique(q >> np.uint64(24)))
    want[T - 1] = True
    if stride:
        want[::stride] = True
    assert (np.flatnonzero(want) == pos).all()


def test_histogram_quantiles():
    q = bench.hist_summary([90, 5, 3, 1, 0, 0, 1])
    assert q["p50"] == 1 and q["p90"] == 1 and q["p99"] == 15 and q["max_bin_upper"] == 127 and q["n"] == 100
