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


def test_headline_line_is_short_whatever_the_detail_holds():
    """the driver's line is built from a fixed set of keys: prose, histograms and per-kernel tables of the full record never reach it"""
    import json
    import bench
    big = "x" * 4000
    leg = dict(workload=big, reads=2_000_000, read_len=150, seq_mode=1, ms_per_step=80.123456, mreads_per_s=25.0, gbp_per_s=3.75, sub_batches=1,
               stage_ms=dict(extract=1.0, sort=2.0, join=3.0, order=0.0, score=4.0, total=10.0), kernel_ms={f"k{i}": dict(ms=1.0, launches=1) for i in range(40)},
               roofline=dict(frac=1.45, note=big), parity=dict(reads=16384, mismatches=0, index=big, dead_matches=dict(note=big)), query_runs=dict(note=big))
    out = dict(metric="Mreads/s classified", value=62.123456789, unit="Mreads/s", n_gpus=1, steps=20, warmup=5, ms_per_step=160.6, higher_is_better=True, scaling="weak",
               vs_baseline=None, dtype="u64", data="synthetic",
               config=dict(workload="10M x 150 bp ...", reads_per_gpu=10_000_000, read_len=150, targets=16_000_000_000, seq_mode=1, gbp_per_s=9.3, note=big),
               stage_ms=dict(extract=13.5, sort=30.7, join=73.7, regroup=0.0, segsort=0.0, score=42.7, total=160.6), kernel_ms=leg["kernel_ms"],
               roofline=dict(bound="hbm", kernel="join", achieved=3270.0, peak=8000.0, unit="GB/s", frac=0.41, traffic=1.527e11, effective=0.26, write_amplification=2.13,
                             avg_launch_ms=74.2, launches=1, algorithmic_bytes_per_launch=2.4268e11, peak_measured=4950.0, note=big, traffic_note=big, footprint=dict(note=big)),
               roofline_all={f"k{i}": dict(ms=1.0, note=big) for i in range(20)}, join_footprint=dict(note=big), run_lengths=dict(index=dict(note=big), queries=dict(note=big)),
               best_case=dict(leg), other_configs=dict(paired=dict(leg), long=dict(leg), novel=dict(leg)),
               cpu_baseline=dict(value=0.118, unit="Mreads/s", cores=256, kind="port", sample="first 1000000 reads ...", cpu_model="AMD EPYC 9575F", single_thread_value=0.0108,
                                 index_targets=1_320_000_000, numa={f"node{i}": big for i in range(8)}, stage_seconds=dict(match=7.0), cold_cache_note=big),
               parity_sample=dict(reads=1_000_000, mismatches=0, matches=125_845_141, oracle_matches=125_845_141, ambiguous_excluded=0, index=big, dead_matches=dict(note=big)),
               ranks=[dict(rank=i, library=big) for i in range(8)], deferred_reads=dict(note=big), ab=None)
    text = json.dumps(bench.headline(out, "bench_detail.json"))
    assert len(text) < 4096 and big[:100] not in text
    line = json.loads(text)
    assert line["roofline"]["frac"] == 0.41 and line["cpu_baseline"]["cores"] == 256 and line["parity_sample"]["mismatches"] == 0
    assert set(line["other_configs"]) == {"paired", "long", "novel", "best_case"} and "roofline" not in line["other_configs"]["long"]
    assert line["detail"] == "bench_detail.json" and line["value"] == 62.12346
