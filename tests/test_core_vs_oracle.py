"""CPU checks of the kernel arithmetic (tests/emu = host build of mtb_core.h /
mtb_score_par.h, the exact functions the HIP kernels call) against the oracle,
and of the oracle against the committed golden vectors."""
import os

import numpy as np
import pytest

from helpers import default_params, tax2species_table, tax_arrays

HERE = os.path.dirname(os.path.abspath(__file__))


def test_extract_equals_oracle(toy, orc, emu):
    ke, el, el2 = emu.extract_batch(toy.p, toy.b1, toy.o1, toy.b2, toy.o2)
    ko, ql, ql2 = orc.extract_batch(toy.p, toy.b1, toy.o1, toy.b2, toy.o2)
    assert (el == ql).all() and (el2 == ql2).all()
    assert len(ke) == len(ko) and (ke == ko).all()


@pytest.mark.parametrize("syncmer", [0, 1])
@pytest.mark.parametrize("smer_len", [3, 5, 7])
def test_extract_edge_cases(orc, emu, syncmer, smer_len):
    rng = np.random.default_rng(1)
    seqs = [b"", b"A", b"ACGTACGTACGTACGTACGTACGTA", b"ACGTACGTACGTACGTACGTACGTAC", b"ACGTACGTACGTACGTACGTACGTACG", b"N" * 60,
            b"acgtnRYKMSWBDHVU" * 8]
    for L in list(range(20, 40)) + [149, 150, 151, 152, 300, 1001]:
        s = rng.choice(np.frombuffer(b"ACGT", np.uint8), size=L)
        if L > 30:
            s[rng.integers(0, L, size=max(1, L // 50))] = ord("N")
        seqs.append(s.tobytes())
    bases = np.frombuffer(b"".join(seqs), dtype=np.uint8).copy()
    offs = np.zeros(len(seqs) + 1, np.uint64)
    offs[1:] = np.cumsum([len(s) for s in seqs])
    p = default_params(seq_mode=1, syncmer=syncmer, smer_len=smer_len)
    ke, el, _ = emu.extract_batch(p, bases, offs)
    ko, ql, _ = orc.extract_batch(p, bases, offs)
    assert (el == ql).all() and len(ke) == len(ko) and (ke == ko).all()
    # mates of different lengths, one of them too short -> the pair is skipped
    p2 = default_params(seq_mode=2, syncmer=syncmer, smer_len=smer_len)
    order = rng.permutation(len(seqs))
    b2 = np.frombuffer(b"".join(seqs[i] for i in order), dtype=np.uint8).copy()
    o2 = np.zeros(len(seqs) + 1, np.uint64)
    o2[1:] = np.cumsum([len(seqs[i]) for i in order])
    ke, el, el2 = emu.extract_batch(p2, bases, offs, b2, o2)
    ko, ql, ql2 = orc.extract_batch(p2, bases, offs, b2, o2)
    assert (el == ql).all() and (el2 == ql2).all() and len(ke) == len(ko) and (ke == ko).all()


def _tables(toy, orc):
    mx = orc.lib.orc_tax_max_id(toy.tax)
    return tax2species_table(orc, toy.tax, toy.taxids, mx), tax_arrays(orc, toy.tax, toy.world.tax)


def test_join_equals_oracle(toy, orc, emu):
    t2s, _ = _tables(toy, orc)
    q = np.sort(toy.ref["kmers"], order=["value"], kind="stable")
    m = emu.join(toy.values, toy.taxids.astype(np.uint32), t2s, 0xFFFFFFFF, toy.p.kmer_format, q)
    ms = emu.sort_matches(m)
    assert len(ms) == len(toy.ref["matches"]) and (ms == toy.ref["matches"]).all()


def test_join_one_bisection_search_equals_reference_loop(toy, orc, emu):
    """The search k_join_dir runs -- one bisection on the whole value; a target equal to the query ends it with the block of equal
    targets as the selection; otherwise the run is found by stepping from the landing place -- selects what the reference's loop over the
    run of the amino-acid part selects (KmerMatcher.cpp:1117-1146: the minimum is 0, the threshold min(2 x 0, 7) = 0, and only equal DNA
    parts have hamming sum 0)."""
    t2s, _ = _tables(toy, orc)
    q = np.sort(toy.ref["kmers"], order=["value"], kind="stable")
    info = toy.taxids.astype(np.uint32)
    m, n_exact = emu.join_one_bisection(toy.values, info, t2s, 0xFFFFFFFF, toy.p.kmer_format, q)
    ref = emu.join(toy.values, info, t2s, 0xFFFFFFFF, toy.p.kmer_format, q)
    assert len(m) == len(ref) and (m == ref).all()            # same matches in the same (query, index) order
    ms = emu.sort_matches(m)
    assert len(ms) == len(toy.ref["matches"]) and (ms == toy.ref["matches"]).all()
    assert n_exact > 0                                        # reads sampled from the genomes: both branches are taken
    assert n_exact < len(q)


def test_join_one_bisection_search_on_long_runs(orc, emu, tmp_path):
    """The same on a database with runs of > 70 candidates and a spread of hamming sums, including runs where several species hold
    the query's own value (a block of equal targets inside a long run) and queries that land past the end of their run."""
    from conftest import HotToy
    t = HotToy(orc, tmp_path / "db", seq_mode=1, n_reads=200)
    assert t.max_run > 64
    mx = orc.lib.orc_tax_max_id(t.tax)
    t2s = tax2species_table(orc, t.tax, t.taxids, mx)
    info = t.taxids.astype(np.uint32)
    q = np.sort(t.ref["kmers"], order=["value"], kind="stable")
    # edge queries: DNA part all-ones / all-zeros of existing amino-acid parts (land at the end / the start of the run), and absent parts
    from helpers import kmer_dt
    extra = np.zeros(6, kmer_dt)
    v = t.values[len(t.values) // 2]
    aam = ~np.uint64(0xFFFFFF)
    extra["value"] = [v | np.uint64(0xFFFFFF), v & aam, t.values[0] & aam, t.values[-2] | np.uint64(0xFFFFFF), np.uint64(0), (v & aam) + np.uint64(1 << 24)]
    extra["qinfo"] = np.uint64(1) << np.uint64(32)
    q = np.sort(np.concatenate([q, extra]), order=["value"], kind="stable")
    m, n_exact = emu.join_one_bisection(t.values, info, t2s, 0xFFFFFFFF, t.p.kmer_format, q)
    ref = emu.join(t.values, info, t2s, 0xFFFFFFFF, t.p.kmer_format, q)
    assert len(m) == len(ref) and (m == ref).all()
    mo = orc.sort_matches(orc.match(t.db, q))
    ms = emu.sort_matches(m)
    assert len(ms) == len(mo) and (ms == mo).all()
    assert n_exact > 0


def test_join_last_index_entry_is_never_a_candidate(toy, orc, emu):
    from helpers import kmer_dt
    t2s, _ = _tables(toy, orc)
    q = np.zeros(2, kmer_dt)
    q["value"] = [toy.values[-2], toy.values[-1]]
    q["qinfo"] = np.uint64(1) << np.uint64(32)
    me = emu.sort_matches(emu.join(toy.values, toy.taxids.astype(np.uint32), t2s, 0xFFFFFFFF, toy.p.kmer_format, q))
    mo = orc.sort_matches(orc.match(toy.db, q))
    assert len(me) == len(mo) and (me == mo).all()
    # a query equal to the last entry alone finds nothing unless earlier entries share its amino-acid part
    q1 = q[1:].copy()
    m1 = emu.join(toy.values, toy.taxids.astype(np.uint32), t2s, 0xFFFFFFFF, toy.p.kmer_format, q1)
    aam = ~np.uint64(0xFFFFFF)
    n_same = int(((toy.values[:-1] & aam) == (toy.values[-1] & aam)).sum())
    assert (len(m1) == 0) == (n_same == 0)
    assert len(m1) == len(orc.match(toy.db, q1))


def _same_results(toy, res, tt, tc):
    ro = toy.ref["results"]
    amb = ro["flag"] != 0
    assert ((res["classification"] == ro["classification"]) | amb).all()
    assert ((res["score"].view(np.uint32) == ro["score"].view(np.uint32)) | amb).all()
    assert ((res["is_classified"] == ro["is_classified"]) | amb).all()
    if not amb.any():
        assert (tt == toy.ref["tc_tax"]).all() and (tc == toy.ref["tc_cnt"]).all()


def test_score_sequential_core_equals_oracle(toy, orc, emu):
    _, ta = _tables(toy, orc)
    res, tt, tc = emu.score(ta, toy.p, toy.ref["matches"], toy.n_reads, toy.ref["qlen"], toy.ref["qlen2"])
    _same_results(toy, res, tt, tc)


def test_score_parallel_phases_equal_oracle(toy, orc, emu):
    """Data-parallel scorer (rank sort + rounds DP + parallel filter), fed with the
    matches of every read in random order, exactly as k_regroup leaves them."""
    _, ta = _tables(toy, orc)
    m = toy.ref["matches"]
    rng = np.random.default_rng(3)
    seqs = (m["qinfo"] >> np.uint64(32)) & np.uint64(0x1FFFFFFF)
    sh = m[np.lexsort((rng.random(len(m)), seqs))]
    res, tt, tc = emu.score_par(ta, toy.p, sh, toy.n_reads, toy.ref["qlen"], toy.ref["qlen2"], presorted=False)
    _same_results(toy, res, tt, tc)
    assert emu.last_chain_reads > 0           # the pointer-doubling path is exercised ...
    res2, tt2, tc2 = emu.score_par(ta, toy.p, sh, toy.n_reads, toy.ref["qlen"], toy.ref["qlen2"], presorted=False, use_chain=False)
    assert emu.last_chain_reads == 0          # ... and equals the round-by-round DP bit for bit
    assert (res2 == res).all() and (tt2 == tt).all() and (tc2 == tc).all()


@pytest.mark.parametrize("kw", [dict(min_score=0.2), dict(min_sp_score=0.9), dict(tie_ratio=0.5), dict(min_cons_cnt=2, min_cons_cnt_euk=3),
                                dict(min_cons_cnt=1), dict(min_score=0.35, min_sp_score=0.6, tie_ratio=0.8), dict(tie_ratio=0.7)])
@pytest.mark.parametrize("regime", ["clean", "noisy"])
def test_score_parameter_variants(orc, emu, tmp_path, kw, regime):
    from conftest import Toy
    # "noisy": species of a genus 4 % apart, 6 % read errors -> low scores and near-ties, so min_score / tie_ratio bite
    t = Toy(orc, tmp_path, syncmer=1, paired=False, seed=8, n_reads=150, **(dict(err=0.06, genus_div=0.04) if regime == "noisy" else {}))
    for k, v in kw.items():
        setattr(t.p, k, v)
    ref = orc.classify(t.db, t.tax, t.p, t.b1, t.o1)
    _, ta = _tables(t, orc)
    t.ref = ref
    res, tt, tc = emu.score(ta, t.p, ref["matches"], t.n_reads, ref["qlen"], ref["qlen2"])
    _same_results(t, res, tt, tc)
    res, tt, tc = emu.score_par(ta, t.p, ref["matches"], t.n_reads, ref["qlen"], ref["qlen2"], presorted=True)
    _same_results(t, res, tt, tc)


def test_host_taxonomy_loader_equals_oracle(toy, orc, emu):
    """host_db.h (libmtb's loader) vs the oracle's taxonomy services."""
    (canon, parent, depth, under, spp, acc), t2s = emu.load_taxonomy(os.path.join(toy.dbdir, "taxonomy"), np.unique(toy.taxids))
    t2s_o, (canon_o, parent_o, depth_o, under_o, spp_o, acc_o) = _tables(toy, orc)
    assert (canon == canon_o).all() and (depth == depth_o).all() and (under == under_o).all() and (spp == spp_o).all() and (acc == acc_o).all()
    assert (parent[canon >= 0] == parent_o[canon >= 0]).all()
    assert (t2s == t2s_o).all()


GOLDEN = ["toy_sync_se", "toy_dense_pe", "toy_oldfmt_pe", "toy_sync_long"]


def golden_params(g):
    seq_mode = int(g["seq_mode"]) if "seq_mode" in g.files else (2 if int(g["paired"]) else 1)
    kf = int(g["kmer_format"]) if "kmer_format" in g.files else 2
    return default_params(seq_mode=seq_mode, syncmer=int(g["syncmer"]), kmer_format=kf)


@pytest.mark.parametrize("name", GOLDEN)
def test_oracle_reproduces_golden_vectors(orc, tmp_path, name):
    """Regression pin of the oracle on committed vectors (tests/golden/make_golden.py)."""
    from metabuli_amd import synth
    g = np.load(os.path.join(HERE, "golden", name + ".npz"))
    p = golden_params(g)
    tax = synth.Taxonomy()
    for (t, par), r, nm in zip(g["tax_nodes"], g["tax_ranks"], g["tax_names"]):
        tax.add(int(t), int(par), str(r), str(nm))
    d = str(tmp_path)
    tax.write(os.path.join(d, "taxonomy"))
    orc.write_db(d, g["db_values"], g["db_taxids"], p)
    assert (np.fromfile(os.path.join(d, "diffIdx"), dtype=np.uint16) == g["diffidx"]).all()
    t = orc.load_taxonomy(os.path.join(d, "taxonomy"))
    db = orc.open_db(d, t, p)
    paired = bool(int(g["paired"]))
    R = orc.classify(db, t, p, g["bases"], g["offs"], g["bases2"] if paired else None, g["offs2"] if paired else None)
    assert (R["kmers"] == g["kmers"]).all() and (R["matches"] == g["matches"]).all()
    assert (R["results"] == g["results"]).all() and (R["tc_tax"] == g["tc_tax"]).all() and (R["tc_cnt"] == g["tc_cnt"]).all()


def test_slot16_roundtrip(toy, emu):
    """the 16-byte slot form used inside per-read segments keeps every field of a match of a short read"""
    import ctypes as C
    from helpers import match_dt
    m = toy.ref["matches"]
    if len(m) == 0 or (m["qinfo"] & np.uint64(0xFFFFFFFF)).max() >= 4096:
        pytest.skip("positions beyond the slot form (long reads use exact segments)")
    out = np.zeros(len(m), match_dt)
    for epoch in (1, 17, 31):
        rc = emu.lib.emu_slot_roundtrip(m.ctypes.data_as(C.c_void_p), C.c_size_t(len(m)), C.c_uint32(epoch), out.ctypes.data_as(C.c_void_p))
        assert rc == 0
        assert (out == m).all()


def test_oracle_openmp_layer_equals_the_serial_restatement(orc, tmp_path):
    """bench.py times the oracle on all host cores: read chunks, query splits and read blocks in parallel,
    __gnu_parallel::sort for the two sorts.  Same answers as with one thread (match / metamer lists as sorted sets)."""
    from conftest import Toy
    t = Toy(orc, tmp_path, syncmer=1, paired=True, seed=21, n_reads=1500)
    a = t.ref
    b = orc.classify(t.db, t.tax, t.p, t.b1, t.o1, t.b2, t.o2, threads=4)
    assert (np.sort(a["kmers"], order=["value", "qinfo"]) == np.sort(b["kmers"], order=["value", "qinfo"])).all()
    assert (a["matches"] == b["matches"]).all()
    assert (a["results"] == b["results"]).all() and (a["tc_tax"] == b["tc_tax"]).all() and (a["tc_cnt"] == b["tc_cnt"]).all()
    assert orc.lib.orc_get_threads() == 1


def test_oracle_whole_batch_call_equals_the_staged_calls(toy, orc):
    """orc_classify_batch (what bench.py times) = the five staged oracle calls"""
    r = orc.classify_batch(toy.db, toy.tax, toy.p, toy.b1, toy.o1, toy.b2, toy.o2, threads=3)
    assert (r["results"] == toy.ref["results"]).all() and (r["tc_tax"] == toy.ref["tc_tax"]).all() and (r["tc_cnt"] == toy.ref["tc_cnt"]).all()
    assert orc.last_counts["matches"] == len(toy.ref["matches"]) and orc.last_counts["kmers"] == len(toy.ref["kmers"])


def test_reporter_restatements_agree(toy, orc, tmp_path):
    """tests/reporter_spec.py (Python, from Reporter.cpp) and the oracle's C++ writer produce the same files"""
    import ctypes as C
    import reporter_spec as rs
    ref = toy.ref
    names = [f"r{i}" for i in range(toy.n_reads)]
    tv = rs.TaxView(toy.world.tax.parent, toy.world.tax.rank, toy.world.tax.name)
    nm = ("\n".join(names) + "\n").encode()
    for lin in (0, 1):
        a = str(tmp_path / f"a{lin}.tsv"); b = str(tmp_path / f"b{lin}.tsv")
        assert orc.lib.orc_write_classifications2(a.encode(), toy.tax, nm, C.c_size_t(toy.n_reads), ref["results"].ctypes.data_as(C.c_void_p),
                                                  ref["tc_tax"].ctypes.data_as(C.c_void_p), ref["tc_cnt"].ctypes.data_as(C.c_void_p), C.c_int(lin)) == 0
        rs.write_classifications(b, tv, names, ref["results"], ref["tc_tax"], ref["tc_cnt"], lineage=bool(lin))
        assert open(a).read() == open(b).read()
    a = str(tmp_path / "ra.tsv"); b = str(tmp_path / "rb.tsv")
    assert orc.lib.orc_write_report(a.encode(), toy.tax, C.c_size_t(toy.n_reads), ref["results"].ctypes.data_as(C.c_void_p)) == 0
    counts = {}
    for c in ref["results"]["classification"].tolist():
        counts[c] = counts.get(c, 0) + 1
    rs.write_report(b, tv, counts, toy.n_reads)
    assert sorted(open(a).read().split("\n")) == sorted(open(b).read().split("\n"))
