"""Third, independent formulation of the unpinnable stages (tests/bruteforce.py: plain Python / numpy evaluators of the
SURVEY 8a formulas, tables from the fixtures pinned to the reference) against BOTH the oracle (oracle/oracle.cpp) and
the host build of the kernel arithmetic (tests/emu): extraction a2-a4, the diffIdx codec a8, the join a10-a11 and the
per-read scoring a13-a18 (chains of consecutive matches enumerated explicitly instead of the forward DP)."""
import os

import numpy as np
import pytest

import bruteforce as bf
from helpers import default_params, match_dt, tax2species_table


@pytest.fixture(scope="module")
def T():
    return bf.ref_tables()


def _first_reads(toy, n_max, base_budget=60000):
    n = 0
    while n < min(n_max, toy.n_reads) and int(toy.o1[n + 1]) <= base_budget:
        n += 1
    return max(n, 1)


def test_extraction_formula(toy, T):
    if toy.p.kmer_format != 2:
        pytest.skip("the window formula below is kmer_format 2 (OldMetamerScanner geometry is covered by the oracle / emu pair)")
    n = _first_reads(toy, 80)
    v, q = bf.extract_spec(T, toy.b1, toy.o1, toy.b2, toy.o2, syncmer=toy.p.syncmer, smer_len=toy.p.smer_len, reads=range(n))
    k = toy.ref["kmers"]
    seq = (k["qinfo"] >> np.uint64(32)) & np.uint64(0x1FFFFFFF)
    ko = k[seq <= np.uint64(n)]
    got = np.sort(np.rec.fromarrays([v, q], names="value,qinfo"), order=["value", "qinfo"])
    exp = np.sort(ko, order=["value", "qinfo"])
    assert len(got) == len(exp) > 0
    assert (got["value"] == exp["value"]).all() and (got["qinfo"] == exp["qinfo"]).all()


def test_diffidx_formula(toy):
    d16 = np.fromfile(os.path.join(toy.dbdir, "diffIdx"), dtype=np.uint16)
    assert (bf.decode_diffidx(d16) == toy.values).all()


def _spec_matches(toy, T, n):
    k = toy.ref["kmers"]
    seq = (k["qinfo"] >> np.uint64(32)) & np.uint64(0x1FFFFFFF)
    kk = k[seq <= np.uint64(n)]
    d16 = np.fromfile(os.path.join(toy.dbdir, "diffIdx"), dtype=np.uint16)
    values = bf.decode_diffidx(d16)
    info = np.fromfile(os.path.join(toy.dbdir, "info"), dtype=np.uint32)
    m = bf.join_spec(T, values, info, toy.world.tax.species_of, kk["value"], kk["qinfo"], kmer_format=toy.p.kmer_format)
    return bf.sort_matches_spec(m), kk


def _as_records(ms):
    out = np.zeros(len(ms), match_dt)
    for i, (qi, tid, sp, dna, reh, ham) in enumerate(ms):
        out[i] = (qi, tid, sp, dna, reh, ham, 0)
    return out


def test_join_formula(toy, T, orc, emu):
    n = _first_reads(toy, 120, base_budget=40000)
    ms, kk = _spec_matches(toy, T, n)
    got = _as_records(ms)
    mo = toy.ref["matches"]
    seq = (mo["qinfo"] >> np.uint64(32)) & np.uint64(0x1FFFFFFF)
    exp = mo[seq <= np.uint64(n)]
    assert len(got) == len(exp) > 0
    assert (got == exp).all()                                   # vs the oracle's streaming merge
    mx = orc.lib.orc_tax_max_id(toy.tax)
    t2s = tax2species_table(orc, toy.tax, toy.taxids, mx)
    me = emu.sort_matches(emu.join(toy.values, toy.taxids.view(np.uint32), t2s, 0xFFFFFFFF, toy.p.kmer_format, kk))
    assert (got == me).all()                                    # vs the kernel arithmetic


def test_scoring_by_chain_enumeration(toy, T):
    n = _first_reads(toy, 120, base_budget=40000)
    ms, _ = _spec_matches(toy, T, n)
    euk = [t for t, nm in toy.world.tax.name.items() if nm == "Eukaryota"]
    p = toy.p
    sc = bf.ReadScorer(toy.world.tax, p.syncmer, p.smer_len, p.seq_mode, kmer_format=p.kmer_format, min_cons_cnt=p.min_cons_cnt,
                       min_cons_cnt_euk=p.min_cons_cnt_euk, min_score=p.min_score, min_sp_score=p.min_sp_score, tie_ratio=p.tie_ratio,
                       accession_level=p.accession_level, eukaryota=euk[0] if euk else 0)
    by_read = {}
    for m in ms:
        by_read.setdefault((m[0] >> 32) & 0x1FFFFFFF, []).append(m)
    ro = toy.ref["results"]
    checked = classified = 0
    for r in range(1, n + 1):
        if ro["flag"][r - 1]:
            continue
        try:
            cls, score, is_cls, taxcnt = sc.score_read(by_read.get(r, []), int(ro["qlen"][r - 1]), int(ro["qlen2"][r - 1]))
        except bf.TooManyChains:
            continue
        e = ro[r - 1]
        assert cls == e["classification"] and is_cls == e["is_classified"], r
        assert np.float32(score).view(np.uint32) == e["score"].view(np.uint32), r
        a = int(e["taxcnt_off"]); b = a + int(e["n_taxcnt"])
        assert sorted(taxcnt.items()) == list(zip(toy.ref["tc_tax"][a:b].tolist(), toy.ref["tc_cnt"][a:b].tolist())), r
        checked += 1
        classified += is_cls
    assert checked >= min(n, 10) * 0.8 and classified > 0


@pytest.mark.parametrize("kw", [dict(min_score=0.3), dict(min_sp_score=0.5), dict(tie_ratio=0.7), dict(min_cons_cnt=1), dict(min_cons_cnt=6, min_cons_cnt_euk=12)],
                         ids=lambda kw: ",".join(f"{k}={v}" for k, v in kw.items()))
def test_scoring_by_chain_enumeration_with_other_parameters(orc, T, tmp_path, kw):
    """the switches of a14 / a17 in the regime where they bite (close species, noisy reads; same workload as the GPU test)"""
    from conftest import Toy
    t = Toy(orc, tmp_path, syncmer=1, paired=False, seed=8, n_reads=120, err=0.06, genus_div=0.04)
    for k, v in kw.items():
        setattr(t.p, k, v)
    t.ref = orc.classify(t.db, t.tax, t.p, t.b1, t.o1)
    test_scoring_by_chain_enumeration(t, T)
