"""GPU parity tests: every stage of the HIP path, called through the C ABI
(include/mtb.h), against the CPU oracle on the same seeded inputs.  Integer /
byte / index outputs must be bit-exact; the per-read score is compared on its
fp32 bit pattern (tolerance 0 <= the 1e-6 BASELINE.json allows)."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    import metabuli_amd as M
    c = M.Context(0)
    yield c
    c.close()


def _params(toy):
    import metabuli_amd as M
    p = toy.p
    return M.default_params(seq_mode=p.seq_mode, syncmer=p.syncmer, smer_len=p.smer_len, kmer_format=p.kmer_format,
                            accession_level=p.accession_level)


def _sorted_by_value_then_all(k):
    return np.sort(k, order=["value", "qinfo"])


def test_loaded_native_library(ctx):
    import metabuli_amd as M
    assert os.path.exists(M.LIB_PATH)
    assert b"gfx950" in M.lib().mtb_version()


def test_extract_matches_oracle(ctx, toy, orc):
    p = _params(toy)
    k, ql, ql2 = ctx.extract(p, toy.b1, toy.o1, toy.b2, toy.o2)
    ko, qlo, ql2o = orc.extract_batch(toy.p, toy.b1, toy.o1, toy.b2, toy.o2)
    assert (ql == qlo).all() and (ql2 == ql2o).all()
    assert len(k) == len(ko)
    # emission order of the reference: read, mate, frame, window
    assert (k == ko).all()


def test_sort_kmers(ctx, toy):
    k = toy.ref["kmers"]
    rng = np.random.default_rng(0)
    shuffled = k[np.argsort(k["qinfo"], kind="stable")]      # extraction-like order (by read)
    s = ctx.sort_kmers(shuffled)
    assert (np.diff(s["value"].astype(np.uint64)) >= 0).all() if len(s) > 1 else True
    assert (_sorted_by_value_then_all(s) == _sorted_by_value_then_all(k)).all()
    # stable: equal values keep input order (ascending qinfo here)
    same = s["value"][1:] == s["value"][:-1]
    assert (s["qinfo"][1:][same] >= s["qinfo"][:-1][same]).all()
    # idempotent
    assert (ctx.sort_kmers(s) == s).all()
    perm = rng.permutation(len(k))
    s2 = ctx.sort_kmers(k[perm])
    assert (s2["value"] == s["value"]).all()


def test_index_decode(ctx, toy):
    p = _params(toy)
    ix = ctx.open_index(toy.dbdir, p)
    v, info = ix.download()
    assert ix.num_targets == len(toy.values)
    assert (v == toy.values).all()
    assert (info.astype(np.int32) == toy.taxids).all()
    ix.close()


def test_match_and_sort_matches(ctx, toy, orc):
    p = _params(toy)
    ix = ctx.open_index(toy.dbdir, p)
    m = ctx.match(ix, toy.ref["kmers"])
    assert len(m) == len(toy.ref["matches"])
    ms = ctx.sort_matches(m, toy.n_reads)
    assert (ms == toy.ref["matches"]).all()
    # last entry of the index is never a candidate (KmerMatcher.cpp:363,378)
    import metabuli_amd as M
    q = np.zeros(2, M.kmer_dt)
    q["value"] = [toy.values[-2], toy.values[-1]]
    q["qinfo"] = (np.uint64(1) << np.uint64(32))
    mm = ctx.match(ix, q)
    mo = orc.match(toy.db, q)
    assert len(mm) == len(mo)
    ix.close()


def test_score(ctx, toy):
    p = _params(toy)
    ix = ctx.open_index(toy.dbdir, p)
    res, tt, tc = ctx.score(ix, p, toy.ref["matches"], toy.n_reads, toy.ref["qlen"], toy.ref["qlen2"])
    _check_results(toy, res, tt, tc)
    ix.close()


def _check_results(toy, res, tt, tc):
    ro = toy.ref["results"]
    amb = ro["flag"] != 0
    assert ((res["classification"] == ro["classification"]) | amb).all()
    assert ((res["is_classified"] == ro["is_classified"]) | amb).all()
    assert ((res["score"].view(np.uint32) == ro["score"].view(np.uint32)) | amb).all()
    assert (res["qlen"] == ro["qlen"]).all() and (res["qlen2"] == ro["qlen2"]).all()
    assert ((res["n_taxcnt"] == ro["n_taxcnt"]) | amb).all()
    # taxID:match_count lists per read
    for i in range(len(ro)):
        if amb[i]:
            continue
        a = slice(int(res["taxcnt_off"][i]), int(res["taxcnt_off"][i]) + int(res["n_taxcnt"][i]))
        b = slice(int(ro["taxcnt_off"][i]), int(ro["taxcnt_off"][i]) + int(ro["n_taxcnt"][i]))
        assert (tt[a] == toy.ref["tc_tax"][b]).all() and (tc[a] == toy.ref["tc_cnt"][b]).all()


def test_fused_batch(ctx, toy):
    p = _params(toy)
    ix = ctx.open_index(toy.dbdir, p)
    res, tt, tc = ctx.classify_batch(ix, p, toy.b1, toy.o1, toy.b2, toy.o2)
    _check_results(toy, res, tt, tc)
    st = ctx.last_stats()
    assert st.n_reads == toy.n_reads and st.n_kmers == len(toy.ref["kmers"]) and st.n_matches == len(toy.ref["matches"])
    # idempotent / deterministic
    res2, tt2, tc2 = ctx.classify_batch(ix, p, toy.b1, toy.o1, toy.b2, toy.o2)
    assert (res2 == res).all() and (tt2 == tt).all() and (tc2 == tc).all()
    ix.close()


def test_host_taxcnt_arrays_hold_the_packed_lists_only(ctx, toy):
    """The host arrays of mtb_classify_batch receive the taxID:count lists packed: a capacity of exactly the lists' total is enough
    (the device-side slots, one per position bucket of every read, are the library's), one entry less is MTB_ERR_CAPACITY reporting
    the total, and nothing depends on which capacity was offered."""
    p = _params(toy)
    ix = ctx.open_index(toy.dbdir, p)
    res, tt, tc = ctx.classify_batch(ix, p, toy.b1, toy.o1, toy.b2, toy.o2)
    _check_results(toy, res, tt, tc)
    total = int(res["n_taxcnt"].sum())
    assert total > 0
    res2, tt2, tc2 = ctx.classify_batch(ix, p, toy.b1, toy.o1, toy.b2, toy.o2, taxcnt_cap=total)
    assert ctx.last_capacity_retries == 0
    assert (res2 == res).all() and (tt2 == tt).all() and (tc2 == tc).all()
    res3, tt3, tc3 = ctx.classify_batch(ix, p, toy.b1, toy.o1, toy.b2, toy.o2, taxcnt_cap=total - 1)
    assert ctx.last_capacity_retries == 1                    # the second call was given the reported size = total
    assert (res3 == res).all() and (tt3 == tt).all() and (tc3 == tc).all()
    ix.close()


def test_two_bit_reads_give_the_results_of_the_text(ctx, toy):
    """mtb_classify_batch_packed (2-bit codes + invalid mask, what the driver sends over PCIe) = mtb_classify_batch on the text, also
    with invalid letters (N, '-', '*'), IUPAC classes, lower case and lengths that are not multiples of a group of eight"""
    p = _params(toy)
    ix = ctx.open_index(toy.dbdir, p)
    rng = np.random.default_rng(11)
    def spoil(b):
        b = b.copy()
        hit = rng.random(len(b)) < 0.01
        b[hit] = rng.choice(np.frombuffer(b"NnRYKMSWBDHV-*acgt", np.uint8), size=int(hit.sum()))
        return b
    b1 = spoil(toy.b1); b2 = spoil(toy.b2) if toy.b2 is not None else None
    want = ctx.classify_batch(ix, p, b1, toy.o1, b2, toy.o2)
    got = ctx.classify_batch_packed(ix, p, b1, toy.o1, b2, toy.o2)
    for a, b in zip(want, got):
        assert (a == b).all()
    ix.close()


def test_a_prefetched_batch_is_used_by_its_classify_call(ctx, toy):
    """the protocol of mtb.h is prefetch(k+1), classify(k) -- two prefetches are outstanding when classify(k) runs (batch
    k, issued one call earlier, and batch k+1).  With ONE record the library found batch k+1's key in classify(k), threw the prefetch away and
    uploaded every batch a second time.  Here: five ragged batches in protocol order = the same batches one by one, and every prefetch issued
    was consumed by its classify call (mtb_ctx_prefetch_stats)."""
    p = _params(toy)
    ix = ctx.open_index(toy.dbdir, p)
    n = toy.n_reads
    cuts = [0, n // 7, n // 7 + 1, n // 2, n - 3, n]
    o1 = toy.o1.astype(np.int64); o2 = toy.o2.astype(np.int64) if toy.o2 is not None else None

    def part(a, b):
        b1 = toy.b1[o1[a]:o1[b]]; q1 = (o1[a:b + 1] - o1[a]).astype(np.uint64)
        if o2 is None:
            return (b1, q1, None, None)
        return (b1, q1, toy.b2[o2[a]:o2[b]], (o2[a:b + 1] - o2[a]).astype(np.uint64))
    batches = [part(a, b) for a, b in zip(cuts[:-1], cuts[1:])]
    want = [ctx.classify_batch_packed(ix, p, *bt) for bt in batches]
    i0, u0 = ctx.prefetch_stats()
    got = ctx.classify_batches_packed_prefetched(ix, p, batches)
    i1, u1 = ctx.prefetch_stats()
    assert i1 - i0 == len(batches) - 1 and u1 - u0 == len(batches) - 1, (i0, u0, i1, u1)
    for w, g in zip(want, got):
        for a, b in zip(w, g):
            assert len(a) == len(b) and (a == b).all()
    # a prefetch nobody comes for does not disturb the calls after it
    got2 = ctx.classify_batches_packed_prefetched(ix, p, batches[:2])
    _ = ctx.classify_batch_packed(ix, p, *batches[4])
    got3 = ctx.classify_batches_packed_prefetched(ix, p, batches[::-1])
    for w, g in zip(want[::-1], got3):
        for a, b in zip(w, g):
            assert len(a) == len(b) and (a == b).all()
    ix.close()


def test_results_on_their_way_back_while_the_next_batch_runs(ctx, toy):
    """mtb_classify_batch_packed_async: a batch's rows and taxID:count lists are copied out on a download stream while the next batch computes
    (two device-side result buffer sets alternate) and belong to the caller once the NEXT call has returned or after mtb_ctx_wait_results.
    Five ragged batches through the pipeline = the same batches through the synchronous call, one by one; a synchronous call afterwards
    still works; an empty batch in the middle flushes."""
    p = _params(toy)
    ix = ctx.open_index(toy.dbdir, p)
    n = toy.n_reads
    cuts = [0, n // 7, n // 7 + 1, n // 2, n - 3, n]
    o1 = toy.o1.astype(np.int64); o2 = toy.o2.astype(np.int64) if toy.o2 is not None else None

    def part(a, b):
        b1 = toy.b1[o1[a]:o1[b]]; q1 = (o1[a:b + 1] - o1[a]).astype(np.uint64)
        if o2 is None:
            return (b1, q1, None, None)
        return (b1, q1, toy.b2[o2[a]:o2[b]], (o2[a:b + 1] - o2[a]).astype(np.uint64))
    batches = [part(a, b) for a, b in zip(cuts[:-1], cuts[1:])]
    want = [ctx.classify_batch_packed(ix, p, *bt) for bt in batches]
    got = ctx.classify_batches_packed_async(ix, p, batches)
    assert len(got) == len(want)
    for w, g in zip(want, got):
        for a, b in zip(w, g):
            assert len(a) == len(b) and (a == b).all()
    again = ctx.classify_batch_packed(ix, p, *batches[3])
    for a, b in zip(want[3], again):
        assert (a == b).all()
    got2 = ctx.classify_batches_packed_async(ix, p, batches[::-1])
    for w, g in zip(want[::-1], got2):
        for a, b in zip(w, g):
            assert len(a) == len(b) and (a == b).all()
    ix.close()


def test_empty_and_ragged_inputs(ctx, orc):
    import metabuli_amd as M
    from helpers import default_params
    seqs = [b"", b"ACGT", b"ACGTACGTACGTACGTACGTACGTA", b"ACGTACGTACGTACGTACGTACGTAC", b"N" * 40,
            b"acgtRYKMacgtnACGTTTGACCATGGCATTAGCCGATTACAGGCATCGAGGCTAGCTAGGATCGATCGGGATCTAGCTAGC" * 3,
            b"ACGTTTGACCATGGCATTAGCCGATTACAGGCATCGAGGCTAGCTAGGATCGATCGGGATCTAGCTAGCNACGTTTGACCATGGCATTAGCCGATTACAGGCATCGAGG"]
    bases = np.frombuffer(b"".join(seqs), dtype=np.uint8).copy()
    offs = np.zeros(len(seqs) + 1, np.uint64)
    offs[1:] = np.cumsum([len(s) for s in seqs])
    for sync in (0, 1):
        k, ql, _ = ctx.extract(M.default_params(seq_mode=1, syncmer=sync), bases, offs)
        ko, qlo, _ = orc.extract_batch(default_params(seq_mode=1, syncmer=sync), bases, offs)
        assert (ql == qlo).all() and len(k) == len(ko) and (k == ko).all()
    # zero reads
    k, _, _ = ctx.extract(M.default_params(), np.zeros(0, np.uint8), np.zeros(1, np.uint64))
    assert len(k) == 0
    # pairs: a pair is skipped entirely when EITHER mate is too short (KmerExtractor.cpp:443-446, 450-453), the query
    # lengths are still reported; mate-2 positions are shifted by used(L1) + 3
    long_ = b"ACGTTTGACCATGGCATTAGCCGATTACAGGCATCGAGGCTAGCTAGGATCGATCGGGATCTAGCTAGCAACGTTTGACCATGGCATTAGCC"
    m1 = [long_, b"ACGTACGTACGT", long_, long_[:26], long_, b""]
    m2 = [long_[::-1], long_, b"ACGTACGTACGTACGTACGTACGTA", long_, long_[:40], long_]
    def cat(seqs):
        o = np.zeros(len(seqs) + 1, np.uint64); o[1:] = np.cumsum([len(x) for x in seqs])
        return np.frombuffer(b"".join(seqs), dtype=np.uint8).copy(), o
    b1, o1 = cat(m1); b2, o2 = cat(m2)
    for sync in (0, 1):
        k, ql, ql2 = ctx.extract(M.default_params(seq_mode=2, syncmer=sync), b1, o1, b2, o2)
        ko, qlo, ql2o = orc.extract_batch(default_params(seq_mode=2, syncmer=sync), b1, o1, b2, o2)
        assert (ql == qlo).all() and (ql2 == ql2o).all() and len(k) == len(ko) and (k == ko).all()
        seqs = set(((k["qinfo"] >> np.uint64(32)) & np.uint64(0x1FFFFFFF)).tolist())
        assert seqs == {1, 4, 5}                              # pairs 2, 3, 6 have a mate below 26 bases


def test_large_properties(ctx):
    """Size-independent properties at a larger size than the oracle is run on."""
    import metabuli_amd as M
    rng = np.random.default_rng(7)
    n = 3_000_000
    k = np.zeros(n, M.kmer_dt)
    k["value"] = rng.integers(0, 2**63, size=n, dtype=np.uint64) * np.uint64(2) + rng.integers(0, 2, size=n, dtype=np.uint64)
    k["qinfo"] = np.arange(n, dtype=np.uint64)
    s = ctx.sort_kmers(k)
    assert (s["value"][1:] >= s["value"][:-1]).all()
    assert int(s["value"].sum(dtype=np.uint64)) == int(k["value"].sum(dtype=np.uint64))
    assert int(s["qinfo"].sum(dtype=np.uint64)) == int(k["qinfo"].sum(dtype=np.uint64))
    order = np.argsort(k["value"], kind="stable")
    assert (s["qinfo"] == k["qinfo"][order]).all()


def test_classify_driver_tsv_outputs(toy, orc, tmp_path):
    """mtb_classify (host loop over the C ABI, reference command line) writes the same
    _classifications.tsv / _report.tsv as the oracle's restated Reporter."""
    import ctypes as C
    import subprocess
    import metabuli_amd as M
    exe = os.path.join(os.path.dirname(M.LIB_PATH), "mtb_classify")
    if not os.path.exists(exe):
        subprocess.check_call(["make", "-C", os.path.dirname(M.LIB_PATH), "mtb_classify"])
    names = [f"read{i}/x" for i in range(toy.n_reads)]

    def write_fastq(path, bases, offs):
        with open(path, "w") as f:
            for i, nm in enumerate(names):
                s = bytes(bases[int(offs[i]):int(offs[i + 1])]).decode()
                f.write(f"@{nm} some comment\n{s}\n+\n{'I' * len(s)}\n")
    fq1 = str(tmp_path / "r1.fq"); write_fastq(fq1, toy.b1, toy.o1)
    args = [exe, "--seq-mode", str(toy.p.seq_mode), "--max-reads", "97", "--accession-level", str(toy.p.accession_level)]
    args.append(fq1)
    if toy.b2 is not None:
        fq2 = str(tmp_path / "r2.fq"); write_fastq(fq2, toy.b2, toy.o2); args.append(fq2)
    args += [toy.dbdir, str(tmp_path), "job"]
    subprocess.check_call(args, stdout=subprocess.DEVNULL)
    ref = toy.ref
    cpath = str(tmp_path / "oracle_classifications.tsv"); rpath = str(tmp_path / "oracle_report.tsv")
    nm = ("\n".join(names) + "\n").encode()
    assert orc.lib.orc_write_classifications(cpath.encode(), toy.tax, nm, C.c_size_t(toy.n_reads), ref["results"].ctypes.data_as(C.c_void_p),
                                             ref["tc_tax"].ctypes.data_as(C.c_void_p), ref["tc_cnt"].ctypes.data_as(C.c_void_p)) == 0
    assert orc.lib.orc_write_report(rpath.encode(), toy.tax, C.c_size_t(toy.n_reads), ref["results"].ctypes.data_as(C.c_void_p)) == 0
    # the Python restatement of Reporter.cpp (tests/reporter_spec.py) is the primary comparison; the oracle's C++ writer second
    import reporter_spec as rs
    tv = rs.TaxView(toy.world.tax.parent, toy.world.tax.rank, toy.world.tax.name)
    spath = str(tmp_path / "spec_classifications.tsv"); srpath = str(tmp_path / "spec_report.tsv")
    rs.write_classifications(spath, tv, names, ref["results"], ref["tc_tax"], ref["tc_cnt"])
    counts = {}
    for c in ref["results"]["classification"].tolist():
        counts[c] = counts.get(c, 0) + 1
    rs.write_report(srpath, tv, counts, toy.n_reads)
    if not (ref["results"]["flag"] != 0).any():
        got_c = open(str(tmp_path / "job_classifications.tsv")).read(); got_r = open(str(tmp_path / "job_report.tsv")).read()
        assert got_c == open(spath).read()
        assert sorted(got_r.split("\n")) == sorted(open(srpath).read().split("\n"))      # order among equal clade counts is unspecified
        assert got_r.split("\n")[0] == open(srpath).read().split("\n")[0]
        assert got_c == open(cpath).read()
        assert sorted(got_r.split("\n")) == sorted(open(rpath).read().split("\n"))
        krona = open(str(tmp_path / "job_krona.html")).read()
        nodes = rs.krona_nodes(tv, counts, toy.n_reads)
        assert krona.endswith("</krona></div></body></html>") and "<krona" in krona
        body = krona[krona.index('<node name="all">'):-len("</krona></div></body></html>")]
        assert sorted(body.split("<node ")) == sorted(nodes.split("<node "))                 # same nodes; sibling order among ties free
        assert body.count("</node>") == nodes.count("</node>")
    # --async-results 1: every batch's results copied out while the next one computes; the same files
    args_a = args[:1] + ["--async-results", "1"] + args[1:-1] + ["joba"]
    subprocess.check_call(args_a, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    assert open(str(tmp_path / "joba_classifications.tsv")).read() == open(str(tmp_path / "job_classifications.tsv")).read()
    assert open(str(tmp_path / "joba_report.tsv")).read() == open(str(tmp_path / "job_report.tsv")).read()
    # --lineage 1 adds the lineage column (Reporter.cpp:38-40, 57-59, 74-76)
    args_l = args[:1] + ["--lineage", "1"] + args[1:-1] + ["jobl"]
    subprocess.check_call(args_l, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    lpath = str(tmp_path / "oracle_classifications_lineage.tsv")
    assert orc.lib.orc_write_classifications2(lpath.encode(), toy.tax, nm, C.c_size_t(toy.n_reads), ref["results"].ctypes.data_as(C.c_void_p),
                                              ref["tc_tax"].ctypes.data_as(C.c_void_p), ref["tc_cnt"].ctypes.data_as(C.c_void_p), C.c_int(1)) == 0
    if not (ref["results"]["flag"] != 0).any():
        got = open(str(tmp_path / "jobl_classifications.tsv")).read()
        rs.write_classifications(spath, tv, names, ref["results"], ref["tc_tax"], ref["tc_cnt"], lineage=True)
        assert got == open(spath).read()
        assert got == open(lpath).read()
        assert "\tlineage\t" in got.split("\n")[0] and ";s_" in got


def test_duplicate_index_entries(ctx, orc, tmp_path):
    """An index with repeated (value, taxid) entries yields identical match records:
    the scorer's strict-compare rank sort must detect the equal keys and fall back."""
    from conftest import Toy
    from helpers import default_params
    import metabuli_amd as M
    t = Toy(orc, tmp_path / "a", syncmer=1, paired=False, seed=31, n_reads=120)
    rng = np.random.default_rng(1)
    dup = np.sort(rng.choice(len(t.values), size=len(t.values) // 10, replace=False))
    idx = np.sort(np.concatenate([np.arange(len(t.values)), dup]))
    vals, tids = t.values[idx], t.taxids[idx]
    d = str(tmp_path / "dup")
    os.makedirs(d)
    t.world.tax.write(os.path.join(d, "taxonomy"))
    p = default_params(seq_mode=1, syncmer=1)
    orc.write_db(d, vals, tids, p)
    tax = orc.load_taxonomy(os.path.join(d, "taxonomy"))
    db = orc.open_db(d, tax, p)
    ref = orc.classify(db, tax, p, t.b1, t.o1)
    assert len(ref["matches"]) > len(t.ref["matches"])       # duplicates really produce extra matches
    mp = M.default_params(seq_mode=1, syncmer=1)
    ix = ctx.open_index(d, mp)
    res, tt, tc = ctx.classify_batch(ix, mp, t.b1, t.o1)
    ro = ref["results"]
    amb = ro["flag"] != 0
    assert ((res["classification"] == ro["classification"]) | amb).all()
    assert ((res["score"].view(np.uint32) == ro["score"].view(np.uint32)) | amb).all()
    if not amb.any():
        assert (tt == ref["tc_tax"]).all() and (tc == ref["tc_cnt"]).all()
    ix.close()


def test_two_streams_give_identical_results(toy):
    """mtb_ctx_set_streams(2): the batch is cut into two read ranges that run concurrently on
    two streams; per-read results and taxcnt entries must equal the single-stream run."""
    import metabuli_amd as M
    c = M.Context(0)
    p = _params(toy)
    ix = c.open_index(toy.dbdir, p)
    big = 8192 // toy.n_reads + 1          # the split needs >= 4096 reads per stream: tile the toy batch
    b1 = np.tile(toy.b1, big); o1 = np.concatenate([[0], np.cumsum(np.tile(np.diff(toy.o1.astype(np.int64)), big))]).astype(np.uint64)
    b2 = o2 = None
    if toy.b2 is not None:
        b2 = np.tile(toy.b2, big); o2 = np.concatenate([[0], np.cumsum(np.tile(np.diff(toy.o2.astype(np.int64)), big))]).astype(np.uint64)
    if len(o1) - 1 < 8192:
        ix.close(); c.close(); pytest.skip("batch too small to be split")
    r1, t1, c1 = c.classify_batch(ix, p, b1, o1, b2, o2)
    c.set_streams(2)
    r2, t2, c2 = c.classify_batch(ix, p, b1, o1, b2, o2)
    assert (r1 == r2).all() and (t1 == t2).all() and (c1 == c2).all()
    n = toy.n_reads
    ro = toy.ref["results"]
    amb = np.tile(ro["flag"] != 0, big)
    assert ((r2["classification"] == np.tile(ro["classification"], big)) | amb).all()
    ix.close(); c.close()


def test_streams_on_an_unsealed_depth7_index(toy, monkeypatch):
    """with mtb_ctx_set_streams(n >= 2) every lane thread used to see an unpacked depth-7 index and launch the
    in-place, non-idempotent pack itself (pack(pack(v)) loses the value) on the index's stream while its join ran on the lane's.
    The conversion now happens once under the index's state lock, complete before any lane's join may read the array; a lane that
    needs the other state waits for the joins in flight.  Two lanes on a freshly opened (flat) depth-7 toy index, then a stage
    join (unpacks), then two lanes again (re-pack): every result equals the single-stream run and the oracle, and the index still
    downloads as the database's arrays."""
    import metabuli_amd as M
    if toy.p.seq_mode == 3:
        pytest.skip("long reads use exact segments (k_join), not the slot path")
    monkeypatch.setenv("MTB_DIR_DEPTH", "7")
    c = M.Context(0)
    p = _params(toy)
    ix = c.open_index(toy.dbdir, p)
    monkeypatch.delenv("MTB_DIR_DEPTH")
    assert ix.state() == dict(dir_depth=7, packed=False, sealed=False)
    big = 8192 // toy.n_reads + 1
    b1 = np.tile(toy.b1, big); o1 = np.concatenate([[0], np.cumsum(np.tile(np.diff(toy.o1.astype(np.int64)), big))]).astype(np.uint64)
    b2 = o2 = None
    if toy.b2 is not None:
        b2 = np.tile(toy.b2, big); o2 = np.concatenate([[0], np.cumsum(np.tile(np.diff(toy.o2.astype(np.int64)), big))]).astype(np.uint64)
    if len(o1) - 1 < 8192:
        ix.close(); c.close(); pytest.skip("batch too small to be split")
    ro = toy.ref["results"]
    amb = np.tile(ro["flag"] != 0, big)
    c.set_streams(2)
    r2, t2, c2 = c.classify_batch(ix, p, b1, o1, b2, o2)              # both lanes arrive at a flat index: one of them packs, once
    if not ix.state()["packed"]:
        ix.close(); c.close(); pytest.skip("this mode's reads carry more metamers than the slot path takes (exact segments on the flat index)")
    assert ((r2["classification"] == np.tile(ro["classification"], big)) | amb).all()
    assert ((r2["score"].view(np.uint32) == np.tile(ro["score"].view(np.uint32), big)) | amb).all()
    m = c.sort_matches(c.match(ix, toy.ref["kmers"]), toy.n_reads)     # stage join: unpacks
    assert (m == toy.ref["matches"]).all() and not ix.state()["packed"]
    r3, t3, c3 = c.classify_batch(ix, p, b1, o1, b2, o2)              # two lanes again: packs again
    assert (r3 == r2).all() and (t3 == t2).all() and (c3 == c2).all()
    c.set_streams(1)
    r1, t1, c1 = c.classify_batch(ix, p, b1, o1, b2, o2)
    assert (r1["classification"] == r2["classification"]).all() and (r1["score"].view(np.uint32) == r2["score"].view(np.uint32)).all()
    v, info = ix.download()
    assert (v == toy.values).all() and (info.astype(np.int32) == toy.taxids).all()
    ix.close(); c.close()


def test_views_keep_their_parent_flat(toy, monkeypatch):
    """a view (mtb_index_slice) reads the parent's flat arrays; while one is alive the parent is not packed (the
    fused join takes its flat-state kernel) and cannot be sealed; after the last view is closed it packs again."""
    import metabuli_amd as M
    if toy.p.seq_mode == 3:
        pytest.skip("long reads use exact segments (k_join), not the slot path")
    monkeypatch.setenv("MTB_DIR_DEPTH", "7")
    c = M.Context(0)
    p = _params(toy)
    ix = c.open_index(toy.dbdir, p)
    monkeypatch.delenv("MTB_DIR_DEPTH")
    res, tt, tc = c.classify_batch(ix, p, toy.b1, toy.o1, toy.b2, toy.o2)
    _check_results(toy, res, tt, tc)
    if not ix.state()["packed"]:
        ix.close(); c.close(); pytest.skip("this mode's reads carry more metamers than the slot path takes (exact segments on the flat index)")
    mid = int(toy.values[len(toy.values) // 2]) & ~0xFFFFFF
    view = ix.slice(0, mid, False)
    assert not ix.state()["packed"]
    res, tt, tc = c.classify_batch(ix, p, toy.b1, toy.o1, toy.b2, toy.o2)          # parent stays flat: k_join_dir<false>
    _check_results(toy, res, tt, tc)
    assert not ix.state()["packed"]
    with pytest.raises(M.MtbError) as e:
        ix.seal()
    assert e.value.status == M.MTB_ERR_UNSUPPORTED
    ks = toy.ref["kmers"]
    lo_part = ks[ks["value"] < np.uint64(mid)]
    m = c.sort_matches(c.match(view, lo_part), toy.n_reads)                          # the view's arrays are still values, not packed words
    m_parent = c.sort_matches(c.match(ix, lo_part), toy.n_reads)
    assert len(m) > 0 and (m == m_parent).all()
    view.close()
    res, tt, tc = c.classify_batch(ix, p, toy.b1, toy.o1, toy.b2, toy.o2)
    _check_results(toy, res, tt, tc)
    assert ix.state()["packed"]
    ix.close(); c.close()


class _Hip:
    """device buffers through the HIP runtime libmtb has loaded (torch cannot initialise its own after it in the same process)"""

    def __init__(self):
        import ctypes as C
        self.C = C
        if os.environ.get("MTB_HIPEMU"):        # the emulated build (tests/hipemu): "device" memory is host memory, the calls are exported under other names
            class _Names:
                pass
            e = C.CDLL(os.environ["MTB_LIB"])
            self.lib = _Names()
            self.lib.hipMalloc, self.lib.hipMemcpy, self.lib.hipFree, self.lib.hipDeviceSynchronize = e.hipemu_c_malloc, e.hipemu_c_memcpy, e.hipemu_c_free, e.hipemu_c_device_synchronize
            return
        self.lib = C.CDLL("libamdhip64.so.7")

    def to_device(self, a):
        C = self.C
        p = C.c_void_p()
        assert self.lib.hipMalloc(C.byref(p), C.c_size_t(max(a.nbytes, 8))) == 0
        assert self.lib.hipMemcpy(p, a.ctypes.data_as(C.c_void_p), C.c_size_t(a.nbytes), C.c_int(1)) == 0
        return p

    def to_host(self, p, like):
        C = self.C
        out = np.empty_like(like)
        assert self.lib.hipDeviceSynchronize() == 0
        assert self.lib.hipMemcpy(out.ctypes.data_as(C.c_void_p), p, C.c_size_t(out.nbytes), C.c_int(2)) == 0
        return out

    def free(self, p):
        self.lib.hipFree(p)


def test_borrowed_arrays_are_handed_back_flat(ctx, toy, monkeypatch):
    """mtb_index_from_device borrows the caller's arrays and the fused path packs d_values in place; closing the
    index restores them (values and, unless the index was sealed, info)."""
    if toy.p.seq_mode == 3:
        pytest.skip("long reads use exact segments (k_join), not the slot path")
    hip = _Hip()
    vals = np.ascontiguousarray(toy.values, dtype=np.uint64); tids = np.ascontiguousarray(toy.taxids, dtype=np.int32)
    dv, di = hip.to_device(vals), hip.to_device(tids)
    monkeypatch.setenv("MTB_DIR_DEPTH", "7")
    p = _params(toy)
    taxid_list = np.unique(tids).astype(np.int32)
    ix = ctx.index_from_device(dv.value, di.value, len(vals), os.path.join(toy.dbdir, "taxonomy"), taxid_list, p)
    monkeypatch.delenv("MTB_DIR_DEPTH")
    res, tt, tc = ctx.classify_batch(ix, p, toy.b1, toy.o1, toy.b2, toy.o2)
    _check_results(toy, res, tt, tc)
    if ix.state()["packed"]:                                                        # (modes that take the exact-segment path never pack)
        assert not (hip.to_host(dv, vals) == vals).all()                            # the lender's array holds packed words now
    ix.close()
    assert (hip.to_host(dv, vals) == vals).all() and (hip.to_host(di, tids) == tids).all()
    hip.free(dv); hip.free(di)


def test_long_reads_on_ordinal_slots_and_on_exact_segments(toy, monkeypatch):
    """Long reads (seq_mode 3): the directory join writes the first match of a read's ord-th metamer to slot ord of the read's own
    slot range, k_seg_order turns the range into the read's segment in compareMatches order by a stable species partition + a rank
    merge of the tail matches (matches that are alone in their species are dropped: they can never be part of a path), k_score_long
    scores it.  With MTB_NO_LONG_SLOTS the same reads take regroup + segment sort.  Both must equal the oracle; the statistics say
    which way the reads went."""
    import metabuli_amd as M
    if toy.p.seq_mode != 3:
        pytest.skip("long-read modes only")
    c = M.Context(0)
    p = _params(toy)
    ix = c.open_index(toy.dbdir, p)
    res, tt, tc = c.classify_batch(ix, p, toy.b1, toy.o1, toy.b2, toy.o2)
    _check_results(toy, res, tt, tc)
    st = c.last_stats()
    assert st.n_slot_reads == toy.n_reads and st.n_matches == len(toy.ref["matches"])
    assert st.n_generic_reads <= toy.n_reads // 2          # (reads with more position buckets than k_score_long's LDS table take the generic kernel)
    c.set_option("MTB_NO_LONG_SLOTS", "1")          # (the environment is read once, at mtb_ctx_create: a live context is switched through its API)
    res2, tt2, tc2 = c.classify_batch(ix, p, toy.b1, toy.o1, toy.b2, toy.o2)
    _check_results(toy, res2, tt2, tc2)
    st = c.last_stats()
    assert st.n_slot_reads == 0 and st.n_matches == len(toy.ref["matches"])
    assert (res2["classification"] == res["classification"]).all() and (res2["score"].view(np.uint32) == res["score"].view(np.uint32)).all()
    ix.close(); c.close()


def test_foreign_sequence_ids_are_rejected(ctx):
    """caller-supplied match records whose sequenceID lies outside the batch are an argument error, not a wild write"""
    import metabuli_amd as M
    m = np.zeros(3, M.match_dt)
    m["qinfo"] = (np.array([1, 2, 9], np.uint64) << np.uint64(32))
    with pytest.raises(M.MtbError) as e:
        ctx.sort_matches(m, 2)
    assert e.value.status == M.MTB_ERR_ARG
    assert len(ctx.sort_matches(m[:2], 2)) == 2


def test_index_write_reproduces_the_database_files(ctx, toy, tmp_path):
    """mtb_index_write (IndexCreator::writeTargetFilesAndSplits restated on the product side): the files written from
    the resident index are byte-identical to the ones the oracle's writer produced, and the copy classifies alike"""
    import shutil
    p = _params(toy)
    ix = ctx.open_index(toy.dbdir, p)
    out = tmp_path / "copy"
    out.mkdir()
    ix.write(str(out))
    ix.close()
    for name in ("diffIdx", "info", "split", "taxID_list"):
        assert (out / name).read_bytes() == open(os.path.join(toy.dbdir, name), "rb").read(), name
    shutil.copytree(os.path.join(toy.dbdir, "taxonomy"), out / "taxonomy")
    p2 = _params(toy)
    ix2 = ctx.open_index(str(out), p2)
    assert (p2.syncmer, p2.smer_len, p2.kmer_format, p2.accession_level) == (p.syncmer, p.smer_len, p.kmer_format, p.accession_level)
    res, tt, tc = ctx.classify_batch(ix2, p2, toy.b1, toy.o1, toy.b2, toy.o2)
    assert (res["classification"] == toy.ref["results"]["classification"]).all()
    ix2.close()


@pytest.mark.parametrize("chunk,packed", [(16, False), (37, True), (4096, False), (4096, True)])
def test_database_opens_chunk_by_chunk(orc, tmp_path, monkeypatch, chunk, packed):
    """mtb_index_open decodes the diffIdx stream in chunks (here: 16 / 37 / 4096 sixteen-bit words, so that metamers of 1-5 words
    are cut by chunk ends in every way): carried words, the running value, the directory rows built per chunk, and -- packed --
    the info entries folded into packed words chunk by chunk (the index opens sealed).  The arrays download as the database's, the
    batch classifies as the oracle says; the same for value ranges opened through the split checkpoints."""
    import metabuli_amd as M
    from conftest import Toy
    toy = Toy(orc, tmp_path / "db", syncmer=1, paired=False, seed=31, n_reads=120, genome_len=1200)       # ~20 k targets: hundreds to thousands of tiny chunks (a chunk costs ~3 ms of launches and syncs)
    monkeypatch.setenv("MTB_OPEN_CHUNK", str(chunk))
    if packed:
        monkeypatch.setenv("MTB_DIR_DEPTH", "7"); monkeypatch.setenv("MTB_OPEN_PACKED", "1")
    c = M.Context(0)
    p = _params(toy)
    ix = c.open_index(toy.dbdir, p)
    st = ix.open_stats()
    assert st["chunk_words"] == min(chunk, os.path.getsize(os.path.join(toy.dbdir, "diffIdx")) // 2) and st["chunks"] >= len(toy.values) // chunk and st["packed_on_load"] == packed
    if packed:
        assert ix.state() == dict(dir_depth=7, packed=True, sealed=True)
    res, tt, tc = c.classify_batch(ix, p, toy.b1, toy.o1, toy.b2, toy.o2)
    _check_results(toy, res, tt, tc)
    v, info = ix.download()
    assert (v == toy.values).all() and (info.astype(np.int32) == toy.taxids).all()
    ix.close()
    for world in ((3,) if chunk == 4096 else ()):
        vs, infos = [], []
        for r in range(world):
            pp = _params(toy)
            part = c.open_index_part(toy.dbdir, pp, r, world)
            a, b = part.download(); vs.append(a); infos.append(b); part.close()
        assert (np.concatenate(vs) == toy.values).all() and (np.concatenate(infos).astype(np.int32) == toy.taxids).all()
    c.close()


def _bench_script(root):
    """bench.py -- or, when the tests run against the emulated library build (tests/hipemu), the wrapper that hands bench.main() a CPU device"""
    return os.path.join(root, "tests", "hipemu", "bench_emulated.py") if os.environ.get("MTB_HIPEMU") else os.path.join(root, "bench.py")


_needs_device = pytest.mark.skipif(bool(os.environ.get("MTB_HIPEMU")), reason="needs a HIP device of its own (torch device tensors / a hipcc build); not part of the emulated run (tests/hipemu)")


@_needs_device
def test_open_under_a_workspace_limit_smaller_than_the_raw_diffidx(tmp_path):
    """a 40 M-target synthetic database written by the device-side coder (mtb_index_write), opened with a workspace limit of 1/16
    of its diffIdx file: the chunked decode keeps the peak at what the index itself holds plus one small chunk (the whole-file
    decode of round 3 held the raw file, the values, the info entries and scan workspace at once), and the arrays come back equal"""
    import torch
    import metabuli_amd as M
    from metabuli_amd import synth
    dev = torch.device("cuda:0")
    c = M.Context(0)
    w = synth.make_world(seed=3, n_genera=1, species_per_genus=2, strains_per_species=1, genome_len=2000, n_filler_species=500)
    taxdir = str(tmp_path / "tax"); w.tax.write(taxdir)
    p = M.default_params(seq_mode=1, syncmer=1)
    T = 40_000_000
    dv = torch.empty(T, dtype=torch.int64, device=dev); di = torch.empty(T, dtype=torch.int32, device=dev)
    n = c.synth_index(7, T, w.filler_tax_lo, w.filler_tax_hi, np.zeros(0, np.uint64), np.zeros(0, np.int32), dv.data_ptr(), di.data_ptr())
    src = c.index_from_device(dv.data_ptr(), di.data_ptr(), n, taxdir, np.arange(w.filler_tax_lo, w.filler_tax_hi + 1, dtype=np.int32), p)
    d = tmp_path / "db"; d.mkdir()
    import shutil
    shutil.copytree(taxdir, d / "taxonomy")
    src.write(str(d))
    raw = os.path.getsize(d / "diffIdx")
    assert os.path.getsize(d / "info") == 4 * n and raw > 2 * n
    want_v = dv[:n].cpu().numpy().view(np.uint64); want_i = di[:n].cpu().numpy()
    src.close(); del dv, di
    torch.cuda.empty_cache()
    c2 = M.Context(0)
    c2.set_workspace_limit(raw // 16)
    p2 = M.default_params(seq_mode=1, syncmer=1)
    ix = c2.open_index(str(d), p2)
    st = ix.open_stats()
    assert st["chunks"] >= 12 and st["chunk_words"] * 2 <= raw // 16
    resident = n * 12 + 4 * (21 ** ix.state()["dir_depth"] + 1)
    assert st["peak_bytes"] < resident + raw // 4, (st, resident, raw)          # far below resident + raw file
    v, info = ix.download()
    assert (v == want_v).all() and (info.view(np.int32) == want_i).all()
    ix.close(); c2.close(); c.close()


@pytest.mark.parametrize("state", ["flat", "packed", "sealed"])
def test_index_clone_is_an_independent_equal_copy(toy, monkeypatch, state):
    """mtb_index_clone (load once, copy to the other GPUs): a second context -- here on the same device -- gets the
    resident index by device copies in whatever state it is in; the source is closed, and the copy still downloads as the database's
    arrays and classifies as the oracle says"""
    import metabuli_amd as M
    if state != "flat":
        monkeypatch.setenv("MTB_DIR_DEPTH", "7")
    a, b = M.Context(0), M.Context(0)
    p = _params(toy)
    src = a.open_index(toy.dbdir, p)
    monkeypatch.delenv("MTB_DIR_DEPTH", raising=False)
    if state != "flat":
        if toy.p.seq_mode == 3:
            src.close(); a.close(); b.close(); pytest.skip("long toy reads may leave the index flat")
        a.classify_batch(src, p, toy.b1, toy.o1, toy.b2, toy.o2)          # the fused join packs a depth-7 index
        if not src.state()["packed"]:
            src.close(); a.close(); b.close(); pytest.skip("this mode's reads stay on the flat-state path")
    if state == "sealed":
        src.seal()
    st = src.state()
    cp = b.clone_index(src)
    assert cp.state() == st and cp.num_targets == len(toy.values)
    src.close(); a.close()
    res, tt, tc = b.classify_batch(cp, p, toy.b1, toy.o1, toy.b2, toy.o2)
    _check_results(toy, res, tt, tc)
    v, info = cp.download()
    assert (v == toy.values).all() and (info.astype(np.int32) == toy.taxids).all()
    cp.close(); b.close()


def _import_worker(share, dbdir, out_path, b1, o1, b2, o2, pkw):
    """a process of its own: opens the exporter's arrays through the inter-process handles, copies them, classifies on the copy"""
    import metabuli_amd as M
    c = M.Context(0)
    p = M.default_params(**pkw)
    tl = np.loadtxt(os.path.join(dbdir, "taxID_list"), dtype=np.int32, ndmin=1)
    ix = c.import_index(share, os.path.join(dbdir, "taxonomy"), tl, p)
    st = ix.state()                      # (as imported: a mode whose reads take a flat-state path unpacks the array on its first batch)
    res, tt, tc = c.classify_batch(ix, p, b1, o1, b2, o2)
    np.savez(out_path, res=res, tt=tt, tc=tc, packed=st["packed"], sealed=st["sealed"], depth=st["dir_depth"], T=ix.num_targets)
    ix.close(); c.close()


@pytest.mark.parametrize("state,other_process", [("flat", False), ("sealed", False), ("sealed", True)])
def test_resident_index_handed_to_another_context_through_its_share_record(toy, monkeypatch, tmp_path, state, other_process):
    """mtb_index_export / mtb_index_import (one process per GPU: how bench.py --gpus N and torch.distributed launches get the index that rank 0
    built): the exporter describes its resident arrays (inter-process handles, offsets, state), the importer copies them device to device
    into memory of its own and loads the taxonomy itself.  Inside one process the record's pointers are used directly; ACROSS processes
    (spawned worker, real GPU only) the handles are opened.  The import classifies as the oracle says, in the exporter's state."""
    import multiprocessing as mp
    import metabuli_amd as M
    if other_process and (os.environ.get("MTB_HIPEMU") or not os.environ.get("MTB_TEST_IPC")):
        pytest.skip("inter-process handles need the real HIP runtime -- and are opt-in (MTB_TEST_IPC=1): on the test pool hipIpcOpenMemHandle of another "
                    "process's allocation succeeded on one box, answered 'invalid argument' on a second and hung on a third (profiles/r06_notes.md section 3)")
    if state != "flat":
        monkeypatch.setenv("MTB_DIR_DEPTH", "7")
    a = M.Context(0)
    p = _params(toy)
    src = a.open_index(toy.dbdir, p)
    monkeypatch.delenv("MTB_DIR_DEPTH", raising=False)
    if state == "sealed":
        if toy.p.seq_mode == 3:
            src.close(); a.close(); pytest.skip("long toy reads may leave the index flat")
        src.seal()
    share = src.export()
    assert len(share) == M.SHARE_BYTES
    tl = np.loadtxt(os.path.join(toy.dbdir, "taxID_list"), dtype=np.int32, ndmin=1)
    if other_process:
        out = str(tmp_path / "imp.npz")
        ctx = mp.get_context("spawn")
        w = ctx.Process(target=_import_worker, args=(share, toy.dbdir, out, toy.b1, toy.o1, toy.b2, toy.o2,
                                                        dict(seq_mode=toy.p.seq_mode, syncmer=toy.p.syncmer, smer_len=toy.p.smer_len, kmer_format=toy.p.kmer_format, accession_level=toy.p.accession_level)))
        w.start(); w.join(600)
        assert w.exitcode == 0
        z = np.load(out)
        assert bool(z["packed"]) == src.state()["packed"] and int(z["T"]) == len(toy.values)
        _check_results(toy, z["res"], z["tt"], z["tc"])
        src.close(); a.close()
        return
    b = M.Context(0)
    cp = b.import_index(share, os.path.join(toy.dbdir, "taxonomy"), tl, p)
    assert cp.state() == src.state() and cp.num_targets == len(toy.values)
    src.close(); a.close()                                  # the import is independent of its source
    res, tt, tc = b.classify_batch(cp, p, toy.b1, toy.o1, toy.b2, toy.o2)
    _check_results(toy, res, tt, tc)
    v, info = cp.download()
    assert (v == toy.values).all() and (info.astype(np.int32) == toy.taxids).all()
    cp.close(); b.close()


@pytest.mark.parametrize("paired", [False, True])
def test_a_few_long_reads_do_not_send_the_batch_down_the_exact_path(orc, tmp_path, paired):
    """one read of >= 4093 used bases (positions beyond a slot record's 12 bits) or with more than 384 metamers used
    to send the WHOLE short-read batch through a second extraction and the exact-segment path.  Now the extractor marks such reads,
    the join files their matches in the overflow list and only they are scored from exact segments: the batch stays on the slot
    path and every read -- 150 bp, 600 bp, 5 kb, 9 kb -- gets the oracle's answer."""
    import metabuli_amd as M
    from conftest import Toy
    from metabuli_amd import synth
    t = Toy(orc, tmp_path / "db", syncmer=1, paired=paired, seed=41, n_reads=300)
    rng = np.random.default_rng(9)
    # splice longer reads (both mates long for pairs) into the batch at scattered places
    extra_at = {5: 5000, 77: 600, 150: 9000, 151: 700, 299: 4200}
    def rebuild(b, o):
        seqs = [b[int(o[i]):int(o[i + 1])] for i in range(len(o) - 1)]
        for at, L in extra_at.items():
            tid, g = t.world.genomes[at % len(t.world.genomes)]
            st = int(rng.integers(0, len(g) - L))
            seqs[at] = synth.mutate(rng, g[st:st + L], 0.01)
        oo = np.zeros(len(seqs) + 1, np.uint64); oo[1:] = np.cumsum([len(x) for x in seqs])
        return np.concatenate(seqs), oo
    b1, o1 = rebuild(t.b1, t.o1)
    b2, o2 = rebuild(t.b2, t.o2) if paired else (None, None)
    ref = orc.classify(t.db, t.tax, t.p, b1, o1, b2, o2)
    c = M.Context(0)
    p = _params(t)
    ix = c.open_index(t.dbdir, p)
    res, tt, tc = c.classify_batch(ix, p, b1, o1, b2, o2)
    st = c.last_stats()
    assert st.n_slot_reads == 300                                  # the batch kept its slot segments
    ro = ref["results"]; amb = ro["flag"] != 0
    assert ((res["classification"] == ro["classification"]) | amb).all()
    assert ((res["score"].view(np.uint32) == ro["score"].view(np.uint32)) | amb).all()
    assert (res["qlen"] == ro["qlen"]).all() and (res["qlen2"] == ro["qlen2"]).all()
    assert ((res["n_taxcnt"] == ro["n_taxcnt"]) | amb).all()
    assert st.n_matches == len(ref["matches"])
    for at in extra_at:
        assert res["is_classified"][at] == ro["is_classified"][at] == 1
    # a batch made of long reads only still takes the whole-batch route (nothing to keep the slots for)
    sel = sorted(extra_at)
    bl = np.concatenate([b1[int(o1[i]):int(o1[i + 1])] for i in sel]); ol = np.zeros(len(sel) + 1, np.uint64); ol[1:] = np.cumsum([int(o1[i + 1] - o1[i]) for i in sel])
    if not paired:
        r2, _, _ = c.classify_batch(ix, p, bl, ol)
        assert (r2["classification"] == ro["classification"][sel]).all() and c.last_stats().n_slot_reads == 0
    ix.close(); c.close()


def test_slot_epoch_wraps_without_stale_matches(toy, orc):
    """the slot segments of the fused path are never cleared between batches: live slots carry a 5-bit epoch tag that
    wraps every 31 batches.  40 batches on one context, alternating two different read sets, must keep giving the
    oracle's answers (a stale slot read as live would add foreign matches)."""
    import metabuli_amd as M
    if toy.p.seq_mode == 3:
        pytest.skip("long reads use exact segments")
    p = _params(toy)
    ctx = M.Context(0)
    ix = ctx.open_index(toy.dbdir, p)
    n = toy.n_reads
    half = n // 2
    # second read set: the first half of the reads only (same buffers, fewer reads) -> different slot contents
    o1b = toy.o1[:half + 1].copy(); b1b = toy.b1[:int(o1b[-1])]
    o2b = toy.o2[:half + 1].copy() if toy.o2 is not None else None
    b2b = toy.b2[:int(o2b[-1])] if toy.o2 is not None else None
    ref = toy.ref["results"]
    for it in range(40):
        if it % 2 == 0:
            res, tt, tc = ctx.classify_batch(ix, p, toy.b1, toy.o1, toy.b2, toy.o2)
            assert (res["classification"] == ref["classification"]).all(), it
            assert (res["score"].view(np.uint32) == ref["score"].view(np.uint32)).all(), it
            assert (tt == toy.ref["tc_tax"]).all() and (tc == toy.ref["tc_cnt"]).all(), it
        else:
            res, tt, tc = ctx.classify_batch(ix, p, b1b, o1b, b2b, o2b)
            assert (res["classification"] == ref["classification"][:half]).all(), it
            assert (res["score"].view(np.uint32) == ref["score"][:half].view(np.uint32)).all(), it
    ix.close(); ctx.close()


def test_many_matches_per_query_take_the_large_segment_path(ctx, orc, tmp_path):
    """every metamer of one genus is also filed under 20 further species of the same genus: a query then has ~20 matches,
    the tail of the read's slot segment overflows, the read is deferred to the large-segment launch (exact segment from
    live slots + overflow list, sorted in HBM).  Results must still be the oracle's."""
    from helpers import default_params
    from metabuli_amd import synth
    import metabuli_amd as M
    rng = np.random.default_rng(41)
    w = synth.make_world(seed=41, n_genera=2, species_per_genus=2, strains_per_species=1, genome_len=20000, with_euk=False)
    # 20 extra species under genus 0 (taxid 4), each a copy of the first genome's metamers
    nxt = max(w.tax.parent) + 1
    extra_sp = []
    for i in range(20):
        w.tax.add(nxt, 4, "species", f"copy{i}"); extra_sp.append(nxt); nxt += 1
    p = default_params(seq_mode=1, syncmer=1)
    from helpers import build_toy_db
    g0 = w.genomes[0][1]
    k, _, _ = orc.extract_batch(default_params(seq_mode=3, syncmer=1), g0, np.array([0, len(g0)], np.uint64))
    v0 = np.unique(k["value"])
    ev = np.tile(v0, len(extra_sp)); et = np.repeat(np.array(extra_sp, np.int32), len(v0))
    d = str(tmp_path / "db"); os.makedirs(d)
    vals, tids = build_toy_db(orc, w, p, d, extra=(ev, et))
    tax = orc.load_taxonomy(os.path.join(d, "taxonomy"))
    db = orc.open_db(d, tax, p)
    b, o, truth = synth.sample_reads(rng, w, 150, length=150, err=0.01, frac_random=0.1)
    ref = orc.classify(db, tax, p, b, o)
    per_read = np.bincount((ref["matches"]["qinfo"] >> np.uint64(32)).astype(np.int64) & 0x1FFFFFFF, minlength=151)[1:]
    assert per_read.max() > 1000                                   # far beyond the slot segment of a read
    mp = M.default_params(seq_mode=1, syncmer=1)
    ix = ctx.open_index(d, mp)
    res, tt, tc = ctx.classify_batch(ix, mp, b, o)
    ro = ref["results"]
    amb = ro["flag"] != 0
    assert ((res["classification"] == ro["classification"]) | amb).all()
    assert ((res["score"].view(np.uint32) == ro["score"].view(np.uint32)) | amb).all()
    if not amb.any():
        assert (tt == ref["tc_tax"]).all() and (tc == ref["tc_cnt"]).all()
    assert (res["is_classified"] != 0).sum() > 60
    # round 5: such reads (a thousand matches, all of species with pairs of matches) are gathered, sorted in LDS by a workgroup each and
    # scored by the streaming kernel (k_many_sort + k_score_long<2048, 256, 256, 256>), not by the HBM-resident sort any more
    st = ctx.last_stats()
    assert st.n_deferred_reads > 30 and st.n_many_reads > 30, (st.n_deferred_reads, st.n_many_reads)
    assert st.n_matches == len(ref["matches"])
    ix.close()


@pytest.mark.parametrize("seq_mode", [1, 2, 3])
@pytest.mark.parametrize("depth7", [False, True])
def test_long_candidate_runs_are_scanned_by_the_wave(orc, tmp_path, seq_mode, depth7, monkeypatch):
    """runs of > 70 candidates per query (a conserved amino-acid 8-mer filed under 70 species, DNA parts varied): k_join_dir hands
    them to the whole wave (second bisection for the run's end, 64 candidates per step, ballot-ranked emission) -- short reads into
    fixed slot segments, pairs, long reads into slot ranges; on the flat array and on packed words under a depth-7 directory.  The
    per-read answers and the match totals are the oracle's."""
    import metabuli_amd as M
    from conftest import HotToy
    t = HotToy(orc, tmp_path / "db", seq_mode=seq_mode, n_reads=12 if seq_mode == 3 else 150, length=4000 if seq_mode == 3 else 150,
               lognormal=seq_mode == 3, err=0.03 if seq_mode == 3 else 0.01)
    assert t.max_run > 64
    if depth7:
        monkeypatch.setenv("MTB_DIR_DEPTH", "7")
    c = M.Context(0)
    p = M.default_params(seq_mode=seq_mode, syncmer=1)
    ix = c.open_index(t.dbdir, p)
    monkeypatch.delenv("MTB_DIR_DEPTH", raising=False)
    assert ix.state()["dir_depth"] == (7 if depth7 else ix.state()["dir_depth"]) and ix.state()["dir_depth"] > 0
    res, tt, tc = c.classify_batch(ix, p, t.b1, t.o1, t.b2, t.o2)
    if seq_mode != 3:                      # (long reads whose tails overflow twice are redone on exact segments: the flat-state k_join)
        assert ix.state()["packed"] == depth7
    ro = t.ref["results"]
    amb = ro["flag"] != 0
    assert ((res["classification"] == ro["classification"]) | amb).all()
    assert ((res["score"].view(np.uint32) == ro["score"].view(np.uint32)) | amb).all()
    assert ((res["n_taxcnt"] == ro["n_taxcnt"]) | amb).all()
    if not amb.any():
        assert (tt == t.ref["tc_tax"]).all() and (tc == t.ref["tc_cnt"]).all()
    if seq_mode != 3:                      # (the long-read slot path drops matches that are alone in their species before it counts)
        assert c.last_stats().n_matches == len(t.ref["matches"])
    assert (res["is_classified"] != 0).sum() >= (6 if seq_mode == 3 else 60)
    # the stage join (k_join: LDS window) on the same database gives the same match list as the oracle: the two joins agree through it
    m = c.sort_matches(c.match(ix, t.ref["kmers"]), t.n_reads)
    assert (m == t.ref["matches"]).all()
    ix.close(); c.close()


@pytest.mark.parametrize("shape", ["more species than the wave's table", "more survivors than the workgroup sorts"])
def test_deferred_reads_beyond_a_tier_fall_to_the_next(orc, tmp_path, shape):
    """The three tiers of the reads the slot scorers defer (kernels_score_many.h): `k_score_many` (a wave: <= 768 species, <= 192 survivors),
    `k_many_sort` + `k_score_long<2048, 256, 256, 256>` (a workgroup: <= 3072 species, <= 4096 survivors), exact segments sorted in HBM (anything).
    (a) a conserved protein filed sparsely under 2600 species: a read meets more species than the wave's table holds and is handed to the
    workgroup kernel; (b) filed under 680 species in full: a read brings more than 4096 matches that all survive the dead-species drop and
    goes on to the exact-segment path.  Results = the oracle's either way; the statistics say which way the reads went."""
    import metabuli_amd as M
    from conftest import HotToy
    if shape.startswith("more species"):
        t = HotToy(orc, tmp_path / "db", seq_mode=1, n_reads=120, n_hot=2600, keep=0.06)
    else:
        t = HotToy(orc, tmp_path / "db", seq_mode=1, n_reads=120, n_hot=680)
    per_read = np.bincount((t.ref["matches"]["qinfo"] >> np.uint64(32)).astype(np.int64) & 0x1FFFFFFF, minlength=t.n_reads + 1)[1:]
    c = M.Context(0)
    p = M.default_params(seq_mode=1, syncmer=1)
    ix = c.open_index(t.dbdir, p)
    res, tt, tc = c.classify_batch(ix, p, t.b1, t.o1)
    ro = t.ref["results"]
    amb = ro["flag"] != 0
    assert ((res["classification"] == ro["classification"]) | amb).all()
    assert ((res["score"].view(np.uint32) == ro["score"].view(np.uint32)) | amb).all()
    assert ((res["n_taxcnt"] == ro["n_taxcnt"]) | amb).all()
    if not amb.any():
        assert (tt == t.ref["tc_tax"]).all() and (tc == t.ref["tc_cnt"]).all()
    st = c.last_stats()
    assert st.n_matches == len(t.ref["matches"]) and st.n_deferred_reads > 10
    if shape.startswith("more species"):
        sp = t.ref["matches"]["species_id"]; rd = (t.ref["matches"]["qinfo"] >> np.uint64(32)).astype(np.int64) & 0x1FFFFFFF
        n_sp = np.array([len(np.unique(sp[rd == r + 1])) for r in np.flatnonzero(per_read > 700)[:20]])
        assert len(n_sp) and n_sp.max() > 768, n_sp                      # beyond the 1024-entry table's 3/4
        assert st.n_many_reads == st.n_deferred_reads                    # ... and the workgroup kernel took them
    else:
        assert per_read.max() > 4096
        assert st.n_many_reads < st.n_deferred_reads                     # some reads went all the way to the exact segments
    # the same with the tiers' temporaries carved out of the dead metamer / digit buffers (MTB_SCRATCH_ALIAS=1: whatever their size)
    c.set_option("MTB_SCRATCH_ALIAS", "1")
    res2, tt2, tc2 = c.classify_batch(ix, p, t.b1, t.o1)
    if shape.startswith("more species"):               # (the other shape's lists -- > 4096 matches per read -- are larger than anything the join left dead: buffers of their own)
        assert c.last_scratch_bytes > 0
    assert (res2["classification"] == res["classification"]).all() and (res2["score"].view(np.uint32) == res["score"].view(np.uint32)).all()
    assert (tt2 == tt).all() and (tc2 == tc).all()
    st2 = c.last_stats()
    assert (st2.n_matches, st2.n_deferred_reads, st2.n_many_reads) == (st.n_matches, st.n_deferred_reads, st.n_many_reads)
    ix.close(); c.close()


@pytest.mark.parametrize("seq_mode", [1, 3])
def test_queries_that_share_a_long_run_walk_it_in_lockstep(orc, tmp_path, seq_mode, monkeypatch):
    """k_join_dir: sorted queries that meet the SAME long candidate run sit in neighbouring lanes.  High coverage of the genome that carries
    the hot metamers (every hot metamer is met by several reads) with read errors (queries without an equal target): the per-read
    answers and the match totals are the oracle's -- short reads into fixed slot segments (packed words, LDS window forced on and off),
    long reads into slot ranges.  (Written for the lockstep walk of shared runs, an experiment that is compiled out because it measured
    slower -- profiles/experiments/join_lockstep_walk.patch; with that patch and -DMTB_JOIN_LOCKSTEP_MIN=3 on the emulated build it exercises that walk, by default
    the wave scan of many neighbouring queries with one run.)"""
    import metabuli_amd as M
    from conftest import HotToy
    t = HotToy(orc, tmp_path / "db", seq_mode=seq_mode, n_reads=60 if seq_mode == 3 else 3000, length=4000 if seq_mode == 3 else 150,
               lognormal=False, err=0.02, n_hot=90)
    monkeypatch.setenv("MTB_DIR_DEPTH", "7")
    c = M.Context(0)
    p = M.default_params(seq_mode=seq_mode, syncmer=1)
    ix = c.open_index(t.dbdir, p)
    monkeypatch.delenv("MTB_DIR_DEPTH", raising=False)
    ro = t.ref["results"]
    amb = ro["flag"] != 0
    for win in ("1", "0"):           # (long reads, round 6: the window form exists for their slot ranges too -- k_join_dir<true, 1, 1, 8, true>)
        c.set_option("MTB_JOIN_WIN", win); c.set_option("MTB_JOIN_WIN_QT", "48" if win == "1" else None)
        res, tt, tc = c.classify_batch(ix, p, t.b1, t.o1, t.b2, t.o2)
        st = c.last_stats()
        if win == "1" and ix.state()["packed"]:
            assert M.JOIN_VARIANTS[st.join_variant] == "window" and st.join_tiles_windowed > 0 and st.join_tiles_outside == 0, (seq_mode, st.join_variant, st.join_tiles_windowed)
        assert ((res["classification"] == ro["classification"]) | amb).all(), win
        assert ((res["score"].view(np.uint32) == ro["score"].view(np.uint32)) | amb).all(), win
        assert ((res["n_taxcnt"] == ro["n_taxcnt"]) | amb).all(), win
        if not amb.any():
            assert (tt == t.ref["tc_tax"]).all() and (tc == t.ref["tc_cnt"]).all(), win
        if seq_mode != 3:
            assert c.last_stats().n_matches == len(t.ref["matches"]), win
    ix.close(); c.close()


@pytest.mark.parametrize("seq_mode", [1, 2])
def test_target_windows_staged_in_lds_give_the_same_matches(orc, tmp_path, seq_mode, monkeypatch):
    """k_join_dir<.., WIN>: a workgroup.s tile of sorted queries stages the low 32 bits of the target words between its first and its last bucket
    in LDS (bounded before the launch by k_join_tile_win, 4-byte direct-to-LDS loads) and searches / evaluates there; the full word of a selected
    candidate comes from global memory; tiles whose span exceeds the LDS capacity read global memory (same code).  Forced on (MTB_JOIN_WIN=1) with
    tiles of 5, 17, 64 and 256 queries -- spans from a few targets to far beyond the capacity, long candidate runs (the wave scan) inside and
    outside a window -- the per-read answers and the match totals are the oracle's, and equal the sector-random variant's (MTB_JOIN_WIN=0)."""
    import metabuli_amd as M
    from conftest import HotToy
    t = HotToy(orc, tmp_path / "db", seq_mode=seq_mode, n_reads=150)
    monkeypatch.setenv("MTB_DIR_DEPTH", "7")
    c = M.Context(0)
    p = M.default_params(seq_mode=seq_mode, syncmer=1)
    ix = c.open_index(t.dbdir, p)
    monkeypatch.delenv("MTB_DIR_DEPTH", raising=False)
    ro = t.ref["results"]
    amb = ro["flag"] != 0
    windowed = {}
    for win, qt, variant in (("1", "5", None), ("1", "64", None), ("1", "256", None), ("0", "256", None), ("1", "64", "windoww6"), ("1", "17", "windoww7"), ("1", "256", "windoww5")):
        # (windoww<W>: the window form compiled for fewer waves per SIMD -- the A/B instantiations)
        c.set_option("MTB_JOIN_WIN", win); c.set_option("MTB_JOIN_WIN_QT", qt); c.set_option("MTB_JOIN_VARIANT", variant)
        res, tt, tc = c.classify_batch(ix, p, t.b1, t.o1, t.b2, t.o2)
        st = c.last_stats()
        tag = (win, qt, M.JOIN_VARIANTS[st.join_variant])
        assert ix.state()["packed"]
        assert M.JOIN_VARIANTS[st.join_variant] == (variant or ("window" if win == "1" else "q1w6")) and st.join_tuned == 0, tag          # pinned: the tuner stays out
        assert ((res["classification"] == ro["classification"]) | amb).all(), tag
        assert ((res["score"].view(np.uint32) == ro["score"].view(np.uint32)) | amb).all(), tag
        assert ((res["n_taxcnt"] == ro["n_taxcnt"]) | amb).all(), tag
        if not amb.any():
            assert (tt == t.ref["tc_tax"]).all() and (tc == t.ref["tc_cnt"]).all(), tag
        assert st.n_matches == len(t.ref["matches"]), tag
        if win == "1":
            # tiles of the announced size; with the windows bounded before the launch (k_join_tile_win) the statistics say how many were staged,
            # and no tile ever finds a query outside its announced window
            assert st.join_tiles == (st.n_kmers + int(qt) - 1) // int(qt) or st.join_tiles >= st.n_kmers // int(qt), tag
            assert st.join_tiles_outside == 0, tag
            if variant is None:
                windowed[qt] = st.join_tiles_windowed
    assert windowed["5"] > 0 and windowed["64"] > 0, windowed           # small tiles: spans that fit LDS
    assert windowed["256"] < (len(t.ref["matches"]) + 255) // 256 + 64   # (wide tiles over a toy index mostly exceed the capacity)
    ix.close(); c.close()


@pytest.mark.parametrize("n_hot", [3, 6, 11, 14])
def test_runs_of_a_dozen_candidates_inside_and_outside_a_window(orc, tmp_path, n_hot, monkeypatch):
    """Candidate runs of 4 .. 15 targets (a metamer filed under n_hot further species, a third of them with the query's own DNA part): the lengths around
    what a window tile reads ahead (four words before / after the landing place, a run's first four candidates) and around the wave-scan threshold (12) --
    blocks of equal targets longer than four, runs that continue beyond the words read ahead on either side, per-lane evaluation of more than four
    candidates.  Window form with tiles of 9 and 256 queries, the sector-random form, and the thresholds 4 / 64 (every such run scanned by the wave / by
    its lane): the per-read answers and the match totals are the oracle's every time."""
    import metabuli_amd as M
    from conftest import HotToy
    t = HotToy(orc, tmp_path / "db", seq_mode=1, n_reads=150, n_hot=n_hot)
    assert n_hot < t.max_run <= n_hot + 6
    monkeypatch.setenv("MTB_DIR_DEPTH", "7")
    c = M.Context(0)
    p = M.default_params(seq_mode=1, syncmer=1)
    ix = c.open_index(t.dbdir, p)
    monkeypatch.delenv("MTB_DIR_DEPTH", raising=False)
    ro = t.ref["results"]
    amb = ro["flag"] != 0
    for win, qt, coop in (("1", "9", None), ("1", "256", None), ("0", "256", None), ("1", "9", "4"), ("1", "64", "64"), ("0", "256", "4")):
        c.set_option("MTB_JOIN_WIN", win); c.set_option("MTB_JOIN_WIN_QT", qt); c.set_option("MTB_JOIN_COOP_MIN", coop)
        res, tt, tc = c.classify_batch(ix, p, t.b1, t.o1)
        st = c.last_stats()
        tag = (n_hot, win, qt, coop, M.JOIN_VARIANTS[st.join_variant])
        assert ix.state()["packed"]
        assert ((res["classification"] == ro["classification"]) | amb).all(), tag
        assert ((res["score"].view(np.uint32) == ro["score"].view(np.uint32)) | amb).all(), tag
        assert ((res["n_taxcnt"] == ro["n_taxcnt"]) | amb).all(), tag
        if not amb.any():
            assert (tt == t.ref["tc_tax"]).all() and (tc == t.ref["tc_cnt"]).all(), tag
        assert st.n_matches == len(t.ref["matches"]), tag
        if win == "1":
            assert st.join_tiles_outside == 0, tag
            if qt == "9":
                assert st.join_tiles_windowed > 0, tag
    ix.close(); c.close()


@pytest.mark.parametrize("seq_mode", [1, 2])
def test_reads_that_meet_many_species_are_scored_from_their_slots(orc, tmp_path, seq_mode, monkeypatch):
    """A conserved protein filed under 160 species, each of which holds only a sparse subset of its metamers: a read of that gene brings a
    few hundred matches, most of them alone in their species.  Its tail overflows, so the slot scorers defer it; k_score_many
    (kernels_score_many.h) takes it straight from its slots + its entries of the overflow list, drops the species that have no (species,
    frame) group of two matches (Taxonomer.cpp:342: they can never score) and scores the rest in LDS.  Results = the oracle's; the
    statistics say that the kernel ran and that it dropped matches; the exact-segment path of round 4 (MTB_NO_SCORE_MANY) gives the
    same answers."""
    import metabuli_amd as M
    from conftest import HotToy
    t = HotToy(orc, tmp_path / "db", seq_mode=seq_mode, n_reads=200, n_hot=160, keep=0.06)
    c = M.Context(0)
    p = M.default_params(seq_mode=seq_mode, syncmer=1)
    ix = c.open_index(t.dbdir, p)
    ro = t.ref["results"]
    amb = ro["flag"] != 0

    def check(res, tt, tc):
        assert ((res["classification"] == ro["classification"]) | amb).all()
        assert ((res["score"].view(np.uint32) == ro["score"].view(np.uint32)) | amb).all()
        assert ((res["n_taxcnt"] == ro["n_taxcnt"]) | amb).all()
        if not amb.any():
            assert (tt == t.ref["tc_tax"]).all() and (tc == t.ref["tc_cnt"]).all()
        assert c.last_stats().n_matches == len(t.ref["matches"])
    check(*c.classify_batch(ix, p, t.b1, t.o1, t.b2, t.o2))
    st = c.last_stats()
    assert st.n_deferred_reads > 20 and st.n_many_reads > 20, (st.n_deferred_reads, st.n_many_reads)
    assert 0 < st.n_many_kept < st.n_many_matches, (st.n_many_kept, st.n_many_matches)
    c.set_option("MTB_NO_SCORE_MANY", "1")
    check(*c.classify_batch(ix, p, t.b1, t.o1, t.b2, t.o2))
    st2 = c.last_stats()
    assert st2.n_many_reads == 0 and st2.n_deferred_reads == st.n_deferred_reads
    c.set_option("MTB_NO_SCORE_MANY", None)
    check(*c.classify_batch(ix, p, t.b1, t.o1, t.b2, t.o2))
    # the grouped overflow list and the deferred reads' segments inside the buffers the join left dead (the unsorted metamer buffer, the digit arrays;
    # toy buffers are below the 64 MiB at which the library does that by itself): same answers on all three deferred paths, and the carving is reported
    assert c.last_scratch_bytes == 0
    c.set_option("MTB_SCRATCH_ALIAS", "1")
    for sw in (None, "MTB_NO_MANY_SORT", "MTB_NO_SCORE_MANY"):
        if sw:
            c.set_option(sw, "1")
        check(*c.classify_batch(ix, p, t.b1, t.o1, t.b2, t.o2))
        assert c.last_scratch_bytes > 0, sw
        if sw:
            c.set_option(sw, None)
    c.set_option("MTB_SCRATCH_ALIAS", "-1")
    check(*c.classify_batch(ix, p, t.b1, t.o1, t.b2, t.o2))
    assert c.last_scratch_bytes == 0
    ix.close(); c.close()


@pytest.mark.parametrize("depth7", [False, True])
def test_runs_beyond_256_candidates_take_the_second_walk(orc, tmp_path, depth7, monkeypatch):
    """runs of > 330 candidates, all within a few codon changes of the query: every lane of the wave meets more than the four possible
    candidates the one-pass scan keeps in registers, so the run is walked a second time for the emission (k_join_dir: the
    `__any(n_c > 4)` branch) -- found untested by a line-coverage run of the emulated build (profiles/r04_emulated_run_summary.md)."""
    import metabuli_amd as M
    from conftest import HotToy
    t = HotToy(orc, tmp_path / "db", seq_mode=1, n_reads=120, n_hot=330)
    assert t.max_run > 320
    if depth7:
        monkeypatch.setenv("MTB_DIR_DEPTH", "7")
    c = M.Context(0)
    p = M.default_params(seq_mode=1, syncmer=1)
    ix = c.open_index(t.dbdir, p)
    monkeypatch.delenv("MTB_DIR_DEPTH", raising=False)
    res, tt, tc = c.classify_batch(ix, p, t.b1, t.o1)
    ro = t.ref["results"]
    amb = ro["flag"] != 0
    assert ((res["classification"] == ro["classification"]) | amb).all()
    assert ((res["score"].view(np.uint32) == ro["score"].view(np.uint32)) | amb).all()
    assert ((res["n_taxcnt"] == ro["n_taxcnt"]) | amb).all()
    if not amb.any():
        assert (tt == t.ref["tc_tax"]).all() and (tc == t.ref["tc_cnt"]).all()
    assert c.last_stats().n_matches == len(t.ref["matches"])
    ix.close(); c.close()


@pytest.mark.parametrize("name", ["toy_sync_se", "toy_dense_pe", "toy_oldfmt_pe", "toy_sync_long"])
def test_golden_vectors_through_the_c_abi(ctx, orc, tmp_path, name):
    """the committed golden vectors (tests/golden/*.npz: inputs, database arrays and the oracle's outputs at the time
    they were generated) reproduced by the HIP path: sorted query metamers, sorted matches, per-read results"""
    import metabuli_amd as M
    from metabuli_amd import synth
    from test_core_vs_oracle import golden_params
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", name + ".npz"))
    op = golden_params(g)
    tax = synth.Taxonomy()
    for (t, par), r, nm in zip(g["tax_nodes"], g["tax_ranks"], g["tax_names"]):
        tax.add(int(t), int(par), str(r), str(nm))
    d = str(tmp_path)
    tax.write(os.path.join(d, "taxonomy"))
    orc.write_db(d, g["db_values"], g["db_taxids"], op)            # the database files in the reference's format
    p = M.default_params(seq_mode=op.seq_mode, syncmer=op.syncmer, kmer_format=op.kmer_format)
    ix = ctx.open_index(d, p)
    paired = bool(int(g["paired"]))
    b2, o2 = (g["bases2"], g["offs2"]) if paired else (None, None)
    k, ql, ql2 = ctx.extract(p, g["bases"], g["offs"], b2, o2)
    ks = ctx.sort_kmers(k)
    assert (_sorted_by_value_then_all(ks) == _sorted_by_value_then_all(g["kmers"])).all()
    m = ctx.sort_matches(ctx.match(ix, g["kmers"]), len(g["offs"]) - 1)
    assert (m == g["matches"]).all()
    res, tt, tc = ctx.classify_batch(ix, p, g["bases"], g["offs"], b2, o2)
    amb = g["results"]["flag"] != 0
    assert ((res["classification"] == g["results"]["classification"]) | amb).all()
    assert ((res["score"].view(np.uint32) == g["results"]["score"].view(np.uint32)) | amb).all()
    if not amb.any():
        assert (tt == g["tc_tax"]).all() and (tc == g["tc_cnt"]).all()
    ix.close()


VARIANTS = [dict(min_score=0.3), dict(min_sp_score=0.5), dict(tie_ratio=0.7), dict(min_cons_cnt=2, min_cons_cnt_euk=3), dict(min_cons_cnt=1),
            dict(min_score=0.35, min_sp_score=0.6, tie_ratio=0.8), dict(min_cons_cnt=6, min_cons_cnt_euk=12)]


@pytest.mark.parametrize("kw", VARIANTS, ids=lambda kw: ",".join(f"{k}={v}" for k, v in kw.items()))
@pytest.mark.parametrize("paired", [False, True], ids=["se", "pe"])
def test_non_default_scoring_parameters_on_the_hip_path(ctx, orc, tmp_path, kw, paired):
    """--min-score / --min-sp-score / --tie-ratio / --min-cons-cnt(-euk) away from their defaults, through mtb_score (exact
    segments) and the fused batch (slot segments): the min_sp_score branch (Taxonomer.cpp:178-185), the min_score skip
    (:357), tie -> LCA at other ratios, the Eukaryota MIN_DEPTH."""
    from conftest import Toy
    import metabuli_amd as M
    # species of a genus 4 % apart and 6 % read errors: scores spread over 0..0.85 and near-ties between species are common,
    # so every one of these switches changes answers (asserted)
    t = Toy(orc, tmp_path, syncmer=1, paired=paired, seed=8, n_reads=300, err=0.06, genus_div=0.04)
    base = t.ref["results"][["classification", "score", "is_classified"]].copy()
    for k, v in kw.items():
        setattr(t.p, k, v)
    t.ref = orc.classify(t.db, t.tax, t.p, t.b1, t.o1, t.b2, t.o2)
    assert (t.ref["results"][["classification", "score", "is_classified"]] != base).any()
    p = M.default_params(seq_mode=t.p.seq_mode, syncmer=1, **kw)
    ix = ctx.open_index(t.dbdir, p)
    res, tt, tc = ctx.score(ix, p, t.ref["matches"], t.n_reads, t.ref["qlen"], t.ref["qlen2"])
    _check_results(t, res, tt, tc)
    res, tt, tc = ctx.classify_batch(ix, p, t.b1, t.o1, t.b2, t.o2)
    _check_results(t, res, tt, tc)
    ix.close()


def test_redundancy_bit_of_legacy_databases(ctx, orc, tmp_path):
    """Databases written before `Skip_redundancy 1` keep a redundancy flag in bit 31 of every `info` entry; the matcher
    masks it (KmerMatcher.cpp:204-205, 381).  A toy database with the bit set on a third of the entries and
    `Skip_redundancy 0` must classify exactly like its clean twin; without the mask the taxonomy ids would be garbage."""
    from conftest import Toy
    import metabuli_amd as M
    t = Toy(orc, tmp_path / "clean", syncmer=1, paired=False, seed=12, n_reads=300)
    d = str(tmp_path / "legacy"); os.makedirs(d)
    for name in ("diffIdx", "split", "taxID_list"):
        open(os.path.join(d, name), "wb").write(open(os.path.join(t.dbdir, name), "rb").read())
    t.world.tax.write(os.path.join(d, "taxonomy"))
    info = np.fromfile(os.path.join(t.dbdir, "info"), dtype=np.uint32)
    rng = np.random.default_rng(5)
    flagged = rng.random(len(info)) < 0.33
    (info | (flagged.astype(np.uint32) << np.uint32(31))).astype(np.uint32).tofile(os.path.join(d, "info"))
    txt = open(os.path.join(t.dbdir, "db.parameters")).read().replace("Skip_redundancy\t1", "Skip_redundancy\t0")
    assert "Skip_redundancy\t0" in txt
    open(os.path.join(d, "db.parameters"), "w").write(txt)
    # the oracle on the legacy files (skip_redundancy 0 -> mask on) equals the clean run
    from helpers import default_params
    op = default_params(seq_mode=1, syncmer=1, skip_redundancy=0)
    tax = orc.load_taxonomy(os.path.join(d, "taxonomy"))
    ref = orc.classify(orc.open_db(d, tax, op), tax, op, t.b1, t.o1)
    assert (ref["matches"] == t.ref["matches"]).all() and (ref["results"] == t.ref["results"]).all()
    p = M.default_params(seq_mode=1, syncmer=1, skip_redundancy=0)
    ix = ctx.open_index(d, p)
    assert p.skip_redundancy == 0
    _, dl_info = ix.download()
    assert (dl_info >> 31).sum() == flagged.sum()                 # the resident index keeps the raw entries ...
    m = ctx.sort_matches(ctx.match(ix, t.ref["kmers"]), t.n_reads)
    assert (m == t.ref["matches"]).all()                          # ... and the join masks them
    res, tt, tc = ctx.classify_batch(ix, p, t.b1, t.o1)
    _check_results(t, res, tt, tc)
    ix.close()


def test_bench_path_matches_the_oracle(tmp_path):
    """bench.py's own configuration in small: synthetic filler index (mtb_synth_index), borrowed device arrays
    (mtb_index_from_device), device-resident reads (mtb_classify_batch_device) -- the entry points the headline number is
    measured on -- compared with the oracle read by read inside bench.py (`parity_sample`: the sample against the timed index itself,
    the oracle on a sub-database that holds the sample's candidate closure).  A filler-dominated index is a different join /
    scorer regime from the toy genomes: amino-acid runs full of foreign candidates."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for mode in (1, 2):
        out = subprocess.run([sys.executable, _bench_script(root), "--steps", "1", "--warmup", "0", "--reads", "20000", "--targets", "3e6",
                              "--cpu-reads", "20000", "--cpu-stride", "4", "--species", "8", "--genome-len", "150000", "--filler-species", "3000",
                              "--leg-pairs", "3000", "--leg-long", "40", "--leg-long-len", "3000", "--full-parity-reads", "3000",
                              "--seq-mode", str(mode)], capture_output=True, text=True, timeout=900, cwd=str(tmp_path))
        assert out.returncode == 0, out.stderr[-2000:]
        head = json.loads(out.stdout.strip().split("\n")[-1])            # the driver's line: short form
        assert len(out.stdout.strip().split("\n")[-1]) < 8192 and head["parity_sample"]["mismatches"] == 0 and head["cpu_baseline"]["kind"] == "port"
        line = json.load(open(tmp_path / head["detail"]))                 # the full record next to it
        ps = line["parity_sample"]
        assert ps["reads"] == 20000 and ps["mismatches"] == 0 and ps["classified"] > 10000, ps
        assert ps["matches"] == ps["oracle_matches"] > 0
        assert line["cpu_baseline"]["cores"] >= 1 and line["cpu_baseline"]["kind"] == "port"
        assert line["roofline"]["frac"] > 0 and line["run_lengths"]["index"]["runs_by_log2_length"][0] > 0
        if mode == 1:                        # the legs of the other configurations ride on the single-end run
            oc = line["other_configs"]
            assert oc["paired"]["mismatches"] == 0 and oc["long"]["mismatches"] == 0
            assert oc["paired"]["parity"]["reads"] == 3000 and oc["long"]["parity"]["matches"] == oc["long"]["parity"]["oracle_matches"] > 0


@_needs_device
def test_bench_heavy_tailed_workload_in_small(tmp_path):
    """the device-side world generator (conserved protein-coding segments shared at class-dependent prevalence) + shared-run extras:
    200 genomes give candidate runs of > 100 species where the reads hit them; the wave-cooperative scan of k_join_dir answers them
    as the oracle does (headline sample, pairs, long reads), and the best-case leg runs"""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, _bench_script(root), "--steps", "1", "--warmup", "0", "--reads", "100000", "--targets", "2.5e8",
                          "--cpu-reads", "30000", "--cpu-stride", "8", "--species", "200", "--genome-len", "600000", "--filler-species", "5000",
                          "--leg-pairs", "20000", "--leg-long", "100", "--leg-long-len", "5000", "--leg-novel", "30000", "--heldout", "20", "--long-parity-reads", "40", "--full-parity-reads", "4000", "--no-cpu"],
                         capture_output=True, text=True, timeout=1500, cwd=str(tmp_path))
    assert out.returncode == 0, out.stderr[-3000:]
    head = json.loads(out.stdout.strip().split("\n")[-1])
    assert set(head["other_configs"]) == {"paired", "long", "best_case", "novel"} and all(v.get("parity", {"mismatches": 0})["mismatches"] == 0 for v in head["other_configs"].values())
    line = json.load(open(tmp_path / head["detail"]))
    assert line["parity_sample"]["mismatches"] == 0 and line["parity_sample"]["reads"] == 30000
    assert line["other_configs"]["paired"]["mismatches"] == 0 and line["other_configs"]["long"]["mismatches"] == 0
    assert line["other_configs"]["novel"]["mismatches"] == 0 and line["other_configs"]["novel"]["parity"]["reads"] == 4000        # reads of held-out organisms
    assert line["best_case"]["ms_per_step"] > 0 and line["library"]["path"].endswith(".so") and line["ranks"][0]["rank"] == 0
    rl = line["run_lengths"]
    assert rl["index"]["shared_run_extras"] > 0 and rl["index"]["quantiles_over_targets"]["max_bin_upper"] >= 127
    assert rl["queries"]["quantiles"]["max_bin_upper"] >= 127            # queries meet runs of > 64 candidates: the wave-scanned path ran


@_needs_device
def test_bench_two_ranks_share_one_gpu_and_one_index_build(tmp_path):
    """bench.py launched as the driver launches it for N = 2 (torch.distributed.run, one process per rank; here both on cuda:0 over gloo): every rank
    builds its replica of the index and classifies its own reads; one JSON line for the job.  With MTB_TEST_IPC=1 also the hand-over (--handover:
    rank 0 builds and seals the index ONCE, rank 1 imports it through the inter-process handles -- opt-in: see the test above)."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    port = 29800 + os.getpid() % 150
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(port),
                          os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--reads", "200000", "--targets", "6e8", "--species", "200",
                          "--genome-len", "600000", "--filler-species", "5000", "--dist-backend", "gloo", "--shared-gpu"] + (["--handover"] if os.environ.get("MTB_TEST_IPC") else []),
                         capture_output=True, text=True, timeout=600, cwd=str(tmp_path), env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0"))
    assert out.returncode == 0, out.stderr[-3000:]
    head = json.loads([ln for ln in out.stdout.strip().split("\n") if ln.startswith("{")][-1])
    assert head["n_gpus"] == 2 and head["config"]["classified_fraction"] > 0.5
    line = json.load(open(tmp_path / head["detail"]))
    assert len(line["ranks"]) == 2 and line["ranks"][1]["rank"] == 1
    if os.environ.get("MTB_TEST_IPC"):
        assert head["config"]["index_handover"] == "1 of 1 ranks imported rank 0's index", (head["config"]["index_handover"], out.stderr[-2000:])
        assert [r["index"]["mode"] for r in line["ranks"]] == ["exported to the other ranks", "imported from rank 0"]
    else:
        assert [r["index"]["mode"] for r in line["ranks"]] == ["local", "local"]


def test_format1_database_without_kmer_format_line(ctx, orc, tmp_path):
    """setClassifyDefaults (classify.cpp:12) leaves kmerFormat at 1: a legacy database whose db.parameters has no
    Kmer_format line is read as format 1.  mtb_default_params must agree, else such a database is silently extracted
    and joined as format 2."""
    import ctypes as C
    from conftest import Toy
    import metabuli_amd as M
    t = Toy(orc, tmp_path, syncmer=0, paired=False, seed=6, n_reads=200, kmer_format=1)
    path = os.path.join(t.dbdir, "db.parameters")
    lines = [l for l in open(path).read().split("\n") if not l.startswith("Kmer_format")]
    open(path, "w").write("\n".join(lines))
    p = M.Params()
    M.lib().mtb_default_params(C.byref(p))
    assert p.kmer_format == 1 and p.seq_mode == 2 and p.syncmer == 0
    p.seq_mode = 1
    ix = ctx.open_index(t.dbdir, p)
    assert p.kmer_format == 1
    res, tt, tc = ctx.classify_batch(ix, p, t.b1, t.o1)
    _check_results(t, res, tt, tc)
    assert (res["is_classified"] != 0).sum() > 100
    ix.close()


def test_reduced_alphabet_database_is_rejected(ctx, toy, tmp_path):
    """Reduced_alphabet 1 switches the reference to ReducedKmerMatcher and 4-bit codon fields; not implemented here, so
    such a database must be refused instead of being classified with the 24-bit DNA arithmetic"""
    import shutil
    import metabuli_amd as M
    d = str(tmp_path / "red")
    shutil.copytree(toy.dbdir, d)
    path = os.path.join(d, "db.parameters")
    txt = open(path).read().replace("Reduced_alphabet\t0", "Reduced_alphabet\t1")
    assert "Reduced_alphabet\t1" in txt
    open(path, "w").write(txt)
    with pytest.raises(M.MtbError) as e:
        ctx.open_index(d, _params(toy))
    assert e.value.status == M.MTB_ERR_UNSUPPORTED


def test_hbm_budgeted_sub_batches(toy):
    """a21: with a tiny workspace budget the library cuts the batch into >= 3 contiguous read ranges, runs them one after
    another and still returns the undivided batch's results (per-read rows at their places, taxcnt slots appended)."""
    import metabuli_amd as M
    c = M.Context(0)
    p = _params(toy)
    ix = c.open_index(toy.dbdir, p)
    n_bases = int(toy.o1[-1]) + (int(toy.o2[-1]) if toy.o2 is not None else 0)
    c.set_workspace_limit(max(n_bases * 20, 200_000))          # ~a quarter of what the batch needs at once
    res, tt, tc = c.classify_batch(ix, p, toy.b1, toy.o1, toy.b2, toy.o2)
    assert c.last_sub_batches >= 3, c.last_sub_batches
    _check_results(toy, res, tt, tc)
    st = c.last_stats()
    assert st.n_reads == toy.n_reads and st.n_kmers == len(toy.ref["kmers"]) and st.n_matches == len(toy.ref["matches"])
    c.set_workspace_limit(0)
    res2, tt2, tc2 = c.classify_batch(ix, p, toy.b1, toy.o1, toy.b2, toy.o2)
    assert c.last_sub_batches == 1
    assert (res2 == res).all() and (tt2 == tt).all() and (tc2 == tc).all()
    ix.close(); c.close()


def test_packed_index_state_round_trips(toy, monkeypatch, tmp_path):
    """Depth-7 directory + packed target words (kernels_dir.h): the fused join rewrites values[] in place as
    (info << 29 | eighth letter | dna) and reads no info[]; every flat-array user unpacks first.  Forced on a toy index (the
    depth is normally chosen from the index size): fused results == oracle, download == the database's arrays, the stage
    join after a fused batch, the writer after a fused batch, and a second fused batch after all that."""
    import metabuli_amd as M
    if toy.p.seq_mode == 3:
        pytest.skip("long reads use exact segments (k_join), not the slot path")
    monkeypatch.setenv("MTB_DIR_DEPTH", "7")
    c = M.Context(0)
    p = _params(toy)
    ix = c.open_index(toy.dbdir, p)
    monkeypatch.delenv("MTB_DIR_DEPTH")
    res, tt, tc = c.classify_batch(ix, p, toy.b1, toy.o1, toy.b2, toy.o2)          # packs
    _check_results(toy, res, tt, tc)
    ix.seal()                                                                        # info[] released: 8 bytes per target + directory
    res, tt, tc = c.classify_batch(ix, p, toy.b1, toy.o1, toy.b2, toy.o2)
    _check_results(toy, res, tt, tc)
    v, info = ix.download()                                                          # info[] re-allocated, unpacks
    assert (v == toy.values).all() and (info.astype(np.int32) == toy.taxids).all()
    res, tt, tc = c.classify_batch(ix, p, toy.b1, toy.o1, toy.b2, toy.o2)          # packs again
    _check_results(toy, res, tt, tc)
    m = c.sort_matches(c.match(ix, toy.ref["kmers"]), toy.n_reads)                  # stage join: unpacks
    assert (m == toy.ref["matches"]).all()
    res, tt, tc = c.classify_batch(ix, p, toy.b1, toy.o1, toy.b2, toy.o2)
    _check_results(toy, res, tt, tc)
    out = tmp_path / "copy"; out.mkdir()
    ix.write(str(out))                                                               # unpacks
    for name in ("diffIdx", "info"):
        assert (out / name).read_bytes() == open(os.path.join(toy.dbdir, name), "rb").read(), name
    ix.close(); c.close()


@pytest.mark.parametrize("paired,length", [(False, 100), (False, 250), (True, 75), (True, 110), (True, 150)])
def test_register_resident_scorer_variants(ctx, orc, tmp_path, paired, length):
    """k_score_fast instantiations by slot stride: single reads 2 / 3 / 4 elements per lane (<= 128 / 192 / 256 slots), pairs
    3 / 4 / 5 with the (species, frame) runs re-ordered (the slot order of a pair is mate-major, compareMatches order is
    frame-major).  genus_div 0.3 and one strain per species keep most reads on the fast kernel (checked through the
    statistics); every row still has to equal the oracle's."""
    import metabuli_amd as M
    from conftest import Toy
    from helpers import default_params
    from metabuli_amd import synth
    t = Toy.__new__(Toy)
    t.p = default_params(seq_mode=2 if paired else 1, syncmer=1)
    t.world = synth.make_world(seed=40 + length, n_genera=3, species_per_genus=2, strains_per_species=1, genome_len=30000, genus_div=0.3)
    t.dbdir = str(tmp_path)
    from helpers import build_toy_db
    t.values, t.taxids = build_toy_db(orc, t.world, t.p, t.dbdir)
    t.tax = orc.load_taxonomy(os.path.join(t.dbdir, "taxonomy"))
    t.db = orc.open_db(t.dbdir, t.tax, t.p)
    out = synth.sample_reads(np.random.default_rng(length), t.world, 600, length=length, err=0.01, with_n=0.05, paired=paired, lognormal=False)
    if paired:
        t.b1, t.o1, t.b2, t.o2, _ = out
    else:
        (t.b1, t.o1, _), t.b2, t.o2 = out, None, None
    t.n_reads = 600
    t.ref = orc.classify(t.db, t.tax, t.p, t.b1, t.o1, t.b2, t.o2)
    p = M.default_params(seq_mode=t.p.seq_mode, syncmer=1)
    ix = ctx.open_index(t.dbdir, p)
    res, tt, tc = ctx.classify_batch(ix, p, t.b1, t.o1, t.b2, t.o2)
    _check_results(t, res, tt, tc)
    st = ctx.last_stats()
    assert st.n_generic_reads < 0.5 * t.n_reads, st.n_generic_reads          # the fast kernel really took most of them
    assert (res["is_classified"] != 0).sum() > 0.6 * t.n_reads
    ix.close()


@_needs_device
def test_no_kernel_relies_on_zeroed_device_memory(orc, tmp_path):
    """hipMalloc happens to hand out cleared VRAM; nothing may depend on it.  libmtb_xpoison.so (-DMTB_POISON_ALLOC) fills every new
    workspace buffer with 0xA5 on the library's stream before its first use; a single-end and a paired toy batch through the fused
    path (slot segments, register-resident and generic scorer, taxID:count lists) must still equal the oracle -- 34 times in a row after
    an mtb_ctx_reserve that created the slot buffer (the epoch tags of stale slots: see the comment in the child's code)."""
    import subprocess
    import sys
    import metabuli_amd as M
    from conftest import Toy, TOY_MODES
    csrc = os.path.dirname(M.LIB_PATH)
    subprocess.check_call(["make", "-C", csrc, "libmtb_xpoison.so", "X=-DMTB_POISON_ALLOC"], stdout=subprocess.DEVNULL)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for mode in ("sync_se", "sync_pe"):
        t = Toy(orc, tmp_path / mode, **TOY_MODES[mode])
        ref = t.ref["results"]
        exp = str(tmp_path / f"{mode}_expected.npz"); inp = str(tmp_path / f"{mode}_reads.npz")
        np.savez(exp, cls=ref["classification"], score=ref["score"], flag=ref["flag"], tt=t.ref["tc_tax"], tc=t.ref["tc_cnt"])
        np.savez(inp, b1=t.b1, o1=t.o1, **({"b2": t.b2, "o2": t.o2} if t.b2 is not None else {}))
        code = f"""
import sys, numpy as np
sys.path.insert(0, {root!r})
import metabuli_amd as M
assert M.LIB_PATH.endswith("libmtb_xpoison.so"), M.LIB_PATH
r = np.load({inp!r}); e = np.load({exp!r})
c = M.Context(0)
p = M.default_params(seq_mode={int(t.p.seq_mode)}, syncmer=1)
ix = c.open_index({t.dbdir!r}, p)
# ADVICE r4 (high): a slot buffer created by mtb_ctx_reserve was never cleared -- the first batch found "its" pointer and size, skipped the
# memset and ran epoch 1 on whatever the allocation held; with 0xA5 in every byte the stale slots' epoch field reads 20, so batch 20 saw
# every unwritten slot alive.  Reserve first, then more than 20 batches (the tag wraps at 31: 34 batches cross that too).
n_reads = len(r["o1"]) - 1
c.reserve(p, n_reads, int(r["o1"][-1]) * (2 if "b2" in r else 1))
for _ in range(34):
    res, tt, tc = c.classify_batch(ix, p, r["b1"], r["o1"], r["b2"] if "b2" in r else None, r["o2"] if "o2" in r else None)
    amb = e["flag"] != 0
    assert ((res["classification"] == e["cls"]) | amb).all()
    assert ((res["score"].view(np.uint32) == e["score"].view(np.uint32)) | amb).all()
    assert not amb.any() and (tt == e["tt"]).all() and (tc == e["tc"]).all()
print("poisoned run ok", len(res))
"""
        env = dict(os.environ, MTB_LIB=os.path.join(csrc, "libmtb_xpoison.so"))
        p = subprocess.run([sys.executable, "-c", code], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
        assert p.returncode == 0 and "poisoned run ok" in p.stdout, (p.stdout[-500:], p.stderr[-1500:])
