"""GPU tests of the host side above the C ABI: databases that carry only the binary taxonomyDB (internal ids, original ids
in every output), the multi-GPU driver (one engine per device, reads sharded, counts summed), and the C++ stage shims of
include/mtb.hpp executed for real."""
import os
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _exe():
    import metabuli_amd as M
    d = os.path.dirname(M.LIB_PATH)
    subprocess.check_call(["make", "-C", d, "mtb_classify"], stdout=subprocess.DEVNULL)
    return os.path.join(d, "mtb_classify")


def _write_fastq(path, names, bases, offs):
    with open(path, "w") as f:
        for i, nm in enumerate(names):
            s = bytes(bases[int(offs[i]):int(offs[i + 1])]).decode()
            f.write(f"@{nm} comment\n{s}\n+\n{'I' * len(s)}\n")


def _orig_id(t):
    return 1 if t == 1 else 1000 + 7 * t


@pytest.fixture(scope="module")
def twin(orc, tmp_path_factory):
    """toy database A (dump files) and its twin B as the current `build` writes databases: taxa numbered internally
    (TaxonomyWrapper with useInternalTaxID), `info` / taxID_list in internal ids, taxonomy only as binary taxonomyDB"""
    import taxdb_writer as tw
    from conftest import Toy
    from helpers import default_params
    from metabuli_amd import synth
    base = tmp_path_factory.mktemp("twin")
    a = Toy(orc, base / "a", syncmer=1, paired=False, seed=14, n_reads=300)
    tax = a.world.tax
    lines = [(_orig_id(t), _orig_id(tax.parent[t]), tax.rank[t]) for t in sorted(tax.parent)]
    names = {_orig_id(t): tax.name[t] for t in tax.parent}
    bdir = str(base / "b"); os.makedirs(bdir)
    o2i = tw.write_taxonomy_db(os.path.join(bdir, "taxonomyDB"), lines, names, use_internal=True)
    s2i = {t: o2i[_orig_id(t)] for t in tax.parent}              # synth id -> internal id
    tid_i = np.array([s2i[int(t)] for t in a.taxids], np.int32)
    sp_i = np.array([s2i[tax.species_of(int(t))] for t in a.taxids], np.int32)
    order = np.lexsort((tid_i, sp_i, a.values))
    p = default_params(seq_mode=1, syncmer=1)
    orc.write_db(bdir, a.values[order], tid_i[order], p)
    # the same taxonomy in internal ids as dump files, for the oracle only (kept outside the database directory)
    itax = synth.Taxonomy()
    for t in sorted(tax.parent, key=lambda t: s2i[t]):
        itax.add(s2i[t], s2i[tax.parent[t]], tax.rank[t], tax.name[t])
    odir = str(base / "oracle_tax"); itax.write(odir)
    otax = orc.load_taxonomy(odir)
    ref = orc.classify(orc.open_db(bdir, otax, p), otax, p, a.b1, a.o1)
    i2o = {i: o for o, i in o2i.items()}
    return dict(a=a, bdir=bdir, ref=ref, itax=itax, i2o=i2o, s2i=s2i)


def test_database_with_only_taxonomy_db(twin, tmp_path):
    import metabuli_amd as M
    import reporter_spec as rs
    a, ref = twin["a"], twin["ref"]
    assert not os.path.exists(os.path.join(twin["bdir"], "taxonomy"))
    c = M.Context(0)
    p = M.default_params(seq_mode=1, syncmer=1)
    ix = c.open_index(twin["bdir"], p)
    res, tt, tc = c.classify_batch(ix, p, a.b1, a.o1)
    ro = ref["results"]
    assert not (ro["flag"] != 0).any()
    assert (res["classification"] == ro["classification"]).all() and (res["score"].view(np.uint32) == ro["score"].view(np.uint32)).all()
    assert (tt == ref["tc_tax"]).all() and (tc == ref["tc_cnt"]).all()
    assert (res["is_classified"] != 0).sum() > 200
    for i, o in twin["i2o"].items():
        assert ix.original_id(i) == o
    # in original ids the answers are those of the dump-file twin (the renumbering only permutes ids)
    s2i = twin["s2i"]
    assert (np.array([s2i.get(int(t), 0) for t in a.ref["results"]["classification"]]) == res["classification"]).mean() > 0.98
    ix.close(); c.close()
    # the driver prints original ids everywhere (Reporter.cpp:52,62,69,181)
    names = [f"q{i}" for i in range(a.n_reads)]
    fq = str(tmp_path / "r.fq"); _write_fastq(fq, names, a.b1, a.o1)
    subprocess.check_call([_exe(), "--seq-mode", "1", fq, twin["bdir"], str(tmp_path), "job"], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    it = twin["itax"]
    tv = rs.TaxView(it.parent, it.rank, it.name, orig=twin["i2o"] | {0: 0})
    spec_c = str(tmp_path / "spec_c.tsv"); spec_r = str(tmp_path / "spec_r.tsv")
    rs.write_classifications(spec_c, tv, names, ro, ref["tc_tax"], ref["tc_cnt"])
    counts = {}
    for x in ro["classification"].tolist():
        counts[x] = counts.get(x, 0) + 1
    rs.write_report(spec_r, tv, counts, a.n_reads)
    assert open(str(tmp_path / "job_classifications.tsv")).read() == open(spec_c).read()
    assert sorted(open(str(tmp_path / "job_report.tsv")).read().split("\n")) == sorted(open(spec_r).read().split("\n"))
    got = open(str(tmp_path / "job_classifications.tsv")).read()
    assert any(f"\t{_orig_id(t)}\t" in got for t in a.world.species)      # original species ids, not the dense internal ones


def test_outdated_taxonomy_db_falls_back_to_the_dump_files(twin, orc, tmp_path):
    """common.cpp:66-75: an unreadable taxonomyDB ("Outdated taxonomy information") falls through to DBDIR/taxonomy"""
    import shutil
    import taxdb_writer as tw
    import metabuli_amd as M
    a = twin["a"]
    d = str(tmp_path / "db")
    shutil.copytree(a.dbdir, d)
    tax = a.world.tax
    tw.write_taxonomy_db(os.path.join(d, "taxonomyDB"), [(t, tax.parent[t], tax.rank[t]) for t in sorted(tax.parent)], dict(tax.name), version=1)
    c = M.Context(0)
    p = M.default_params(seq_mode=1, syncmer=1)
    ix = c.open_index(d, p)
    res, tt, tc = c.classify_batch(ix, p, a.b1, a.o1)
    assert (res["classification"] == a.ref["results"]["classification"]).all()
    ix.close(); c.close()


@pytest.mark.parametrize("mode", ["sync_se", "sync_pe"])
def test_driver_on_two_engines_equals_one(orc, tmp_path, mode):
    """mtb_classify --devices 0,0: two engines (two contexts + two resident copies of the index, here on the same GPU), every host
    batch cut into two contiguous read ranges classified concurrently from two host threads, rows concatenated in input order,
    per-taxon counts summed.  The files must equal the single-engine run byte for byte."""
    from conftest import Toy, TOY_MODES
    t = Toy(orc, tmp_path / "db", **TOY_MODES[mode])
    names = [f"read{i}" for i in range(t.n_reads)]
    fq1 = str(tmp_path / "r1.fq"); _write_fastq(fq1, names, t.b1, t.o1)
    files = [fq1]
    if t.b2 is not None:
        fq2 = str(tmp_path / "r2.fq"); _write_fastq(fq2, names, t.b2, t.o2); files.append(fq2)
    outs = {}
    for tag, dev in (("one", "0"), ("two", "0,0"), ("three", "0,0,0")):
        od = tmp_path / tag; od.mkdir()
        subprocess.check_call([_exe(), "--seq-mode", str(t.p.seq_mode), "--max-reads", "150", "--devices", dev, "--max-ram", "64", "--match-per-kmer", "8"] + files + [t.dbdir, str(od), "j"],
                              stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        outs[tag] = {f: open(str(od / f"j_{f}")).read() for f in ("classifications.tsv", "report.tsv", "krona.html")}
    assert outs["one"] == outs["two"] == outs["three"]
    assert outs["one"]["classifications.tsv"].count("\n") == t.n_reads + 1


@pytest.mark.parametrize("mode", ["sync_se", "sync_pe"])
def test_driver_on_a_range_partitioned_index_equals_one_engine(orc, tmp_path, mode):
    """mtb_classify --devices 0,0[,0] --partitioned 1: engine d opens only value range d of the database (split checkpoints), every
    host batch goes through mtb_classify_batch_partitioned -- extraction per engine, metamer runs to the range owners and matches
    home as device-to-device peer copies, directory join at the owners, slot scorers at home (SURVEY 8(e) row 2 in the C++ host).
    The files must equal the single-engine run on the whole index byte for byte."""
    from conftest import Toy, TOY_MODES
    t = Toy(orc, tmp_path / "db", **TOY_MODES[mode])
    names = [f"read{i}" for i in range(t.n_reads)]
    fq1 = str(tmp_path / "r1.fq"); _write_fastq(fq1, names, t.b1, t.o1)
    files = [fq1]
    if t.b2 is not None:
        fq2 = str(tmp_path / "r2.fq"); _write_fastq(fq2, names, t.b2, t.o2); files.append(fq2)
    outs = {}
    for tag, extra in (("one", ["--devices", "0"]), ("two", ["--devices", "0,0", "--partitioned", "1"]), ("three", ["--devices", "0,0,0", "--partitioned", "1"])):
        od = tmp_path / tag; od.mkdir()
        p = subprocess.run([_exe(), "--seq-mode", str(t.p.seq_mode), "--max-reads", "150"] + extra + files + [t.dbdir, str(od), "j"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
        assert p.returncode == 0, p.stderr[-2000:]
        outs[tag] = {f: open(str(od / f"j_{f}")).read() for f in ("classifications.tsv", "report.tsv", "krona.html")}
    assert outs["one"] == outs["two"] == outs["three"]
    assert outs["one"]["classifications.tsv"].count("\n") == t.n_reads + 1


STAGE_PROGRAM = r"""
// the reference's loop body (Classifier.cpp:105-119) written against include/mtb.hpp's stage classes, compared with classifyBatch
#include <cstdio>
#include <fstream>
#include "mtb.hpp"
int main(int argc, char **argv) {
    if (argc < 4) return 2;
    mtb_params par; mtb_default_params(&par);
    par.seq_mode = atoi(argv[3]);
    mtb::Engine eng(0, argv[1], "", par);
    mtb::ReadBatch batch;
    std::ifstream in(argv[2]);
    std::string name, s1, s2;
    while (in >> name >> s1) { batch.add(name, s1); if (par.seq_mode == 2) { in >> s2; batch.add_mate(s2); } }
    mtb::KmerExtractor ke(eng, par); mtb::KmerMatcher km(eng); mtb::Classifier cl(eng, par), fused(eng, par);
    mtb::Buffer<mtb::Kmer> kmers(16);                       // deliberately too small: the capacity retry path
    std::vector<mtb::Query> q, q2;
    ke.extractQueryKmers(kmers, q, batch);
    mtb::Buffer<mtb::Match> matches(8);
    int retries = 0;
    while (!km.matchKmers(&kmers, &matches)) { matches.reallocateMemory(km.neededSize()); retries++; }     // Classifier.cpp:127-131
    km.sortMatches(&matches, batch.size());
    cl.assignTaxonomy(matches.buffer, matches.startIndexOfReserve, q);
    fused.classifyBatch(batch, q2);
    if (q.size() != q2.size() || retries != 1) { fprintf(stderr, "size / retry mismatch (%d)\n", retries); return 1; }
    size_t cls = 0;
    for (size_t i = 0; i < q.size(); i++) {
        if (q[i].classification != q2[i].classification || q[i].score != q2[i].score || q[i].isClassified != q2[i].isClassified ||
            q[i].queryLength != q2[i].queryLength || q[i].queryLength2 != q2[i].queryLength2 || q[i].taxCnt != q2[i].taxCnt || q[i].name != q2[i].name) {
            fprintf(stderr, "read %zu differs\n", i); return 1; }
        cls += q[i].isClassified;
        printf("%s\t%d\t%.9g\n", q[i].name.c_str(), q[i].classification, (double)q[i].score);
    }
    if (cl.getTaxCounts() != fused.getTaxCounts()) { fprintf(stderr, "taxCounts differ\n"); return 1; }
    fprintf(stderr, "stages ok: %zu reads, %zu classified, %zu matches\n", q.size(), cls, km.getTotalMatchCnt());
    return 0;
}
"""


@pytest.mark.parametrize("mode", ["sync_se", "dense_pe"])
def test_mtb_hpp_stage_classes_execute(orc, tmp_path, mode):
    """INTEGRATION.md tells a maintainer to link include/mtb.hpp's KmerExtractor / KmerMatcher / Classifier: build a tiny program
    that runs extract -> match (with the `false` + neededSize() retry) -> sortMatches -> assignTaxonomy and compare it with
    classifyBatch inside the program and with the oracle outside."""
    import metabuli_amd as M
    from conftest import Toy, TOY_MODES
    t = Toy(orc, tmp_path / "db", **TOY_MODES[mode])
    src = tmp_path / "stages.cpp"; src.write_text(STAGE_PROGRAM)
    exe = str(tmp_path / "stages")
    libdir = os.path.dirname(M.LIB_PATH)
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-I", os.path.join(ROOT, "include"), "-o", exe, str(src), "-L", libdir, "-lmtb",
                           f"-Wl,-rpath,{libdir}", "-Wl,-rpath,/opt/rocm/lib"])
    reads = tmp_path / "reads.txt"
    with open(reads, "w") as f:
        for i in range(t.n_reads):
            f.write(f"r{i} {bytes(t.b1[int(t.o1[i]):int(t.o1[i + 1])]).decode()}")
            if t.b2 is not None:
                f.write(f" {bytes(t.b2[int(t.o2[i]):int(t.o2[i + 1])]).decode()}")
            f.write("\n")
    out = subprocess.run([exe, t.dbdir, str(reads), str(t.p.seq_mode)], capture_output=True, text=True)
    assert out.returncode == 0, out.stderr
    rows = [l.split("\t") for l in out.stdout.strip().split("\n")]
    ro = t.ref["results"]
    assert len(rows) == t.n_reads
    amb = ro["flag"] != 0
    for i, (nm, cls, sc) in enumerate(rows):
        if amb[i]:
            continue
        assert nm == f"r{i}" and int(cls) == ro["classification"][i] and np.float32(float(sc)).view(np.uint32) == ro["score"][i].view(np.uint32)


@pytest.mark.parametrize("mode", ["sync_se", "sync_pe"])
def test_filter_mode_splits_the_reads(orc, tmp_path, mode):
    """`mtb_classify --filter 1 --print-mode 2` (the `filter` command over the same seam, QueryFilter.cpp:75-118, filter.cpp:5-45):
    classified reads (at the command's default --min-score 0.5) go to <base>_removed.fna, the others to <base>_filtered.fna, as
    ">name\\nsequence\\n" in input order, mates split alike; <base> drops the last extension (two for .gz)."""
    import gzip
    from conftest import Toy, TOY_MODES
    t = Toy(orc, tmp_path / "db", **TOY_MODES[mode])
    names = [f"read{i}" for i in range(t.n_reads)]
    fq1 = str(tmp_path / "sample.R1.fq"); _write_fastq(fq1, names, t.b1, t.o1)
    files = [fq1]
    if t.b2 is not None:
        plain = str(tmp_path / "sample.R2.fq"); _write_fastq(plain, names, t.b2, t.o2)
        fq2 = plain + ".gz"
        with open(plain, "rb") as f, gzip.open(fq2, "wb") as g:
            g.write(f.read())
        os.remove(plain)
        files.append(fq2)
    subprocess.check_call([_exe(), "--filter", "1", "--print-mode", "2", "--seq-mode", str(t.p.seq_mode), "--max-reads", "170"] + files + [t.dbdir],
                          stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    p = type(t.p).from_buffer_copy(t.p); p.min_score = 0.5           # filter.cpp:8
    ref = orc.classify(t.db, t.tax, p, t.b1, t.o1, t.b2, t.o2)["results"]
    cls = ref["is_classified"] != 0
    assert 0 < cls.sum() < t.n_reads                      # both files get reads
    for k, (b, o, path) in enumerate([(t.b1, t.o1, files[0])] + ([(t.b2, t.o2, files[1])] if t.b2 is not None else [])):
        base = str(tmp_path / ("sample.R1" if k == 0 else "sample.R2"))
        def fasta(sel):
            return "".join(f">{names[i]}\n{bytes(b[int(o[i]):int(o[i + 1])]).decode()}\n" for i in range(t.n_reads) if sel[i])
        assert open(base + "_filtered.fna").read() == fasta(~cls)
        assert open(base + "_removed.fna").read() == fasta(cls)
    rows = open(str(tmp_path / "sample.R1_classifications.tsv")).read().split("\n")[1:-1]
    assert [r[0] == "1" for r in rows] == cls.tolist()
    assert os.path.exists(str(tmp_path / "sample.R1_report.tsv")) and not os.path.exists(str(tmp_path / "sample.R1_krona.html"))
