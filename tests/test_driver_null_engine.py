"""The DRIVER (metabuli_amd/csrc/host/classify_main.cpp) on a machine without a GPU: the same source file linked against
tests/null_engine/null_mtb.cpp, a stand-in for libmtb.so's C ABI that classifies nothing -- every read gets a pseudo-result computed from
a checksum of its bases.  What is under test is everything AROUND the engine: the block-parallel parser, batching (--max-reads), the 2-bit
packing of the reads, the cut of a batch over several engines and the rebasing of their taxID:count lists, the order of the rows with
several batches in the GPU stage at once, the prefetch protocol of mtb.h, the exact-size retry, formatting (Reporter.cpp restated in
tests/reporter_spec.py), the report, the filter command's outputs and what happens when a stage fails.

The rule of the null engine is restated here in Python (`predict`), so the expected files are built from the INPUT alone.
The product binary (csrc/mtb_classify) is linked against libmtb.so, which has no CPU path; tests/test_gpu_parity.py runs that one."""
import gzip
import os
import subprocess
import zlib

import numpy as np
import pytest

import reporter_spec as rs
from helpers import result_dt

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NORM = {}
for letters, to in (("ARW", "A"), ("CMS", "C"), ("HTY", "T"), ("BDGKU", "G")):       # common.cpp:13-23 + GeneticCode.h:6 (mtb_core.h: mtb_build_tables)
    for ch in letters:
        NORM[ord(ch)] = to; NORM[ord(ch.lower())] = to
NORM_TABLE = bytes(ord(NORM.get(c, "N")) for c in range(256))


@pytest.fixture(scope="session")
def null_driver(tmp_path_factory):
    d = str(tmp_path_factory.mktemp("null_engine"))
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-shared", "-fPIC", "-o", os.path.join(d, "libmtb_null.so"),
                           os.path.join(ROOT, "tests", "null_engine", "null_mtb.cpp"), "-lz"])
    exe = os.path.join(d, "mtb_classify_null")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-pthread", "-o", exe, os.path.join(ROOT, "metabuli_amd", "csrc", "host", "classify_main.cpp"),
                           "-L" + d, "-l:libmtb_null.so", "-lz", "-Wl,-rpath," + d])
    return exe


@pytest.fixture(scope="session")
def null_db(tmp_path_factory):
    """a database directory as far as the driver and the taxonomy services read it: dump files, taxID_list, db.parameters"""
    from metabuli_amd import synth
    d = str(tmp_path_factory.mktemp("null_db"))
    w = synth.make_world(seed=5, n_genera=3, species_per_genus=3, strains_per_species=2, genome_len=100)
    w.tax.add(max(w.tax.parent) + 1, 2, "species", "Odd <name> & \"quotes\"")
    w.tax.write(os.path.join(d, "taxonomy"))
    ids = sorted(t for t in w.tax.parent if t not in set(w.tax.parent.values()))      # the leaves
    with open(os.path.join(d, "taxID_list"), "w") as f:
        f.write("".join(f"{t}\n" for t in ids))
    with open(os.path.join(d, "db.parameters"), "w") as f:
        f.write("Syncmer\t1\nS-mer_len\t5\nKmer_format\t2\nSkip_redundancy\t1\n")
    np.zeros(1000, np.uint32).tofile(os.path.join(d, "info"))
    return d, w.tax, ids


def predict(reads1, reads2, ids, tc_max=3):
    """the null engine's rule (tests/null_engine/null_mtb.cpp: pseudo), from the texts of the reads"""
    n = len(reads1)
    res = np.zeros(n, result_dt)
    tt, tc = [], []
    for i in range(n):
        text = reads1[i].translate(NORM_TABLE)
        if reads2 is not None:
            text += b"|" + reads2[i].translate(NORM_TABLE)
        h = zlib.crc32(text) | (zlib.crc32(text, 0x5bd1e995) << 32)
        cls = (h & 7) != 0
        res[i]["is_classified"] = cls
        res[i]["classification"] = ids[(h >> 3) % len(ids)] if cls else 0
        res[i]["score"] = np.float32(((h >> 24) % 100001) / 100000.0)
        res[i]["qlen"] = len(reads1[i]); res[i]["qlen2"] = len(reads2[i]) if reads2 is not None else 0
        k_n = 1 + (h >> 44) % tc_max if cls else 0
        res[i]["n_taxcnt"] = k_n; res[i]["taxcnt_off"] = len(tt)
        for k in range(k_n):
            tt.append(ids[((h >> 48) + 7 * k) % len(ids)]); tc.append(1 + ((h >> (52 + 3 * (k % 4))) & 7))
    return res, np.array(tt, np.int32), np.array(tc, np.uint32)


def make_reads(n, seed, lo=30, hi=260, odd=True):
    rng = np.random.default_rng(seed)
    alpha = np.frombuffer(b"ACGT", np.uint8)
    out = []
    for i in range(n):
        L = int(rng.integers(lo, hi))
        s = bytearray(alpha[rng.integers(0, 4, size=L)].tobytes())
        if odd and i % 7 == 3:                    # Ns, lower case, IUPAC codes, other bytes
            for p in rng.integers(0, L, size=5):
                s[int(p)] = int(rng.choice(np.frombuffer(b"NnacgtRYKMSWBDHVU.-*", np.uint8)))
        out.append(bytes(s))
    if odd and n > 12:
        out[5] = b"ACGTACG"; out[6] = b"A"; out[11] = b"N" * 40; out[12] = b"ACGTACGT"      # shorter than a group of 8, exactly one group
    return out


def write_fastq(path, names, reads, gz=False, comment=True):
    data = b"".join(b"@" + nm.encode() + (b" a comment\n" if comment else b"\n") + s + b"\n+\n" + b"I" * len(s) + b"\n" for nm, s in zip(names, reads))
    if gz:
        with gzip.open(path, "wb") as f:
            f.write(data)
    else:
        with open(path, "wb") as f:
            f.write(data)


def write_fasta(path, names, reads, width=0):
    with open(path, "wb") as f:
        for nm, s in zip(names, reads):
            f.write(b">" + nm.encode() + b" desc\n")
            if width:
                for a in range(0, len(s), width):
                    f.write(s[a:a + width] + b"\n")
            else:
                f.write(s + b"\n")


def expected_files(tmp, tax, ids, names, r1, r2, tc_max=3, lineage=False):
    res, tt, tc = predict(r1, r2, ids, tc_max)
    tv = rs.TaxView(tax.parent, tax.rank, tax.name)
    cp, rp = os.path.join(tmp, "expected_classifications.tsv"), os.path.join(tmp, "expected_report.tsv")
    rs.write_classifications(cp, tv, names, res, tt, tc, lineage=lineage)
    counts = {}
    for c in res["classification"].tolist():
        counts[c] = counts.get(c, 0) + 1
    rs.write_report(rp, tv, counts, len(names))
    return open(cp).read(), open(rp).read(), res, counts, tv


def run(exe, args, env=None, check=True):
    e = dict(os.environ)
    e.update(env or {})
    p = subprocess.run([exe] + args, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=e, timeout=300)
    if check:
        assert p.returncode == 0, p.stderr.decode()
    return p


def same_report(got, want):
    g, w = got.split("\n"), want.split("\n")
    return g[0] == w[0] and sorted(g) == sorted(w)          # the order among children of equal clade count is unspecified


@pytest.mark.parametrize("pack", [1, 0])
@pytest.mark.parametrize("extra", [[], ["--devices", "0,1,2"], ["--gpu-workers", "3"], ["--threads", "1"], ["--devices", "1,3", "--gpu-workers", "2"],
                                   ["--async-results", "1"], ["--async-results", "1", "--gpu-workers", "2"]])
def test_single_end_rows_in_input_order(null_driver, null_db, tmp_path, pack, extra):
    d, tax, ids = null_db
    n = 1000
    names = [f"read{i}/1" for i in range(n)]
    r1 = make_reads(n, 1)
    fq = str(tmp_path / "r.fq"); write_fastq(fq, names, r1)
    want_c, want_r, _, _, _ = expected_files(str(tmp_path), tax, ids, names, r1, None)
    p = run(null_driver, ["--seq-mode", "1", "--max-reads", "97", "--pack-reads", str(pack)] + extra + [fq, d, str(tmp_path), "job"], env={"MTB_NULL_VERBOSE": "1"})
    assert open(tmp_path / "job_classifications.tsv").read() == want_c
    assert same_report(open(tmp_path / "job_report.tsv").read(), want_r)
    calls = int(p.stderr.decode().split("null engine: ")[1].split(" classify calls")[0])
    nd = len(extra[1].split(",")) if extra and extra[0] == "--devices" else 1
    assert calls == sum(min(nd, m) for m in [97] * (n // 97) + [n % 97])      # every batch cut into one range per engine
    assert f"The number of processed sequences: {n}" in p.stdout.decode()


@pytest.mark.parametrize("pack", [1, 0])
@pytest.mark.parametrize("kind", ["fastq", "fastq_gz", "fasta", "fasta_wrapped"])
def test_paired_end_and_input_formats(null_driver, null_db, tmp_path, pack, kind):
    d, tax, ids = null_db
    n = 700
    names = [f"pair{i}" for i in range(n)]
    r1, r2 = make_reads(n, 2), make_reads(n, 3)
    if kind.startswith("fastq"):
        gz = kind.endswith("gz")
        f1, f2 = str(tmp_path / ("a.fq.gz" if gz else "a.fq")), str(tmp_path / ("b.fq.gz" if gz else "b.fq"))
        write_fastq(f1, names, r1, gz=gz); write_fastq(f2, names, r2, gz=gz, comment=False)
    else:
        f1, f2 = str(tmp_path / "a.fa"), str(tmp_path / "b.fa")
        w = 60 if kind.endswith("wrapped") else 0
        write_fasta(f1, names, r1, w); write_fasta(f2, names, r2, w)
    want_c, want_r, _, _, _ = expected_files(str(tmp_path), tax, ids, names, r1, r2, lineage=True)
    run(null_driver, ["--seq-mode", "2", "--max-reads", "150", "--pack-reads", str(pack), "--lineage", "1", "--devices", "0,1", f1, f2, d, str(tmp_path), "job"])
    assert open(tmp_path / "job_classifications.tsv").read() == want_c
    assert same_report(open(tmp_path / "job_report.tsv").read(), want_r)


def test_paired_end_with_results_in_flight(null_driver, null_db, tmp_path):
    """--async-results 1 (mtb_classify_batch_packed_async): a batch reaches the formatter after the NEXT call has returned -- the null engine keeps
    the caller's arrays filled with 0xEE until then --, the last one after mtb_ctx_wait_results; one batch only, and no batch at all"""
    d, tax, ids = null_db
    for n, mr in ((700, 64), (50, 1000), (0, 10)):
        names = [f"pair{i}" for i in range(n)]
        r1, r2 = make_reads(n, 20 + n, odd=n > 100), make_reads(n, 21 + n, odd=n > 100)
        f1, f2 = str(tmp_path / "a.fq"), str(tmp_path / "b.fq")
        write_fastq(f1, names, r1); write_fastq(f2, names, r2)
        want_c, want_r, _, _, _ = expected_files(str(tmp_path), tax, ids, names, r1, r2)
        run(null_driver, ["--seq-mode", "2", "--max-reads", str(mr), "--async-results", "1", f1, f2, d, str(tmp_path), "job"])
        assert open(tmp_path / "job_classifications.tsv").read() == want_c
        assert same_report(open(tmp_path / "job_report.tsv").read(), want_r)


def test_taxid_count_lists_longer_than_the_first_guess(null_driver, null_db, tmp_path):
    """the pinned taxID:count arrays start at 6 entries per read; a batch that needs more is redone with the exact size"""
    d, tax, ids = null_db
    n = 7000
    names = [f"r{i}" for i in range(n)]
    r1 = make_reads(n, 4, lo=30, hi=80, odd=False)
    fq = str(tmp_path / "r.fq"); write_fastq(fq, names, r1)
    want_c, want_r, res, _, _ = expected_files(str(tmp_path), tax, ids, names, r1, None, tc_max=16)
    assert res["n_taxcnt"].sum() > 6 * n + 4096
    for extra in ([], ["--devices", "0,1,2"], ["--async-results", "1"]):
        p = run(null_driver, ["--seq-mode", "1", "--max-reads", "10000"] + extra + [fq, d, str(tmp_path), "job"], env={"MTB_NULL_TC_MAX": "16", "MTB_NULL_VERBOSE": "1"})
        assert open(tmp_path / "job_classifications.tsv").read() == want_c
        if extra != ["--devices", "0,1,2"]:
            assert " 1 capacity retries" in p.stderr.decode()


def test_next_batch_is_prefetched(null_driver, null_db, tmp_path):
    """mtb_prefetch_batch_packed: the batch behind the one in work is handed over before that one's classify call, with the arrays its own
    classify call then gets (the null engine refuses anything else)"""
    d, tax, ids = null_db
    n = 1200
    names = [f"r{i}" for i in range(n)]
    r1 = make_reads(n, 5)
    fq = str(tmp_path / "r.fq"); write_fastq(fq, names, r1)
    want_c, _, _, _, _ = expected_files(str(tmp_path), tax, ids, names, r1, None)
    for extra in ([], ["--async-results", "1"]):
        p = run(null_driver, ["--seq-mode", "1", "--max-reads", "100"] + extra + [fq, d, str(tmp_path), "job"], env={"MTB_NULL_DELAY_MS": "30", "MTB_NULL_VERBOSE": "1"})
        assert open(tmp_path / "job_classifications.tsv").read() == want_c
        msg = p.stderr.decode().split("null engine: ")[1]
        n_pref = int(msg.split(" prefetch calls, ")[1].split(" batches found prefetched")[0])
        assert n_pref >= 6, msg                 # 12 batches of 30 ms each: the parser is far ahead after the first ones


def test_engine_failure_ends_the_run(null_driver, null_db, tmp_path):
    d, tax, ids = null_db
    n = 1000
    names = [f"r{i}" for i in range(n)]
    r1 = make_reads(n, 6)
    fq = str(tmp_path / "r.fq"); write_fastq(fq, names, r1)
    want_c, _, _, _, _ = expected_files(str(tmp_path), tax, ids, names, r1, None)
    for extra in ([], ["--gpu-workers", "2"], ["--devices", "0,1"], ["--async-results", "1"], ["--async-results", "1", "--gpu-workers", "2"]):
        p = run(null_driver, ["--seq-mode", "1", "--max-reads", "100"] + extra + [fq, d, str(tmp_path), "job"], env={"MTB_NULL_FAIL_CALL": "4"}, check=False)
        assert p.returncode == 1 and "injected failure" in p.stderr.decode()
        got = open(tmp_path / "job_classifications.tsv").read()
        assert want_c.startswith(got) and len(got) < len(want_c)            # whole batches in order up to the failure, nothing after it
        assert not os.path.exists(tmp_path / "job_report.tsv") or os.path.getmtime(tmp_path / "job_report.tsv") < os.path.getmtime(tmp_path / "job_classifications.tsv")
        if os.path.exists(tmp_path / "job_report.tsv"):
            os.remove(tmp_path / "job_report.tsv")


def test_input_errors(null_driver, null_db, tmp_path):
    d, tax, ids = null_db
    names = [f"r{i}" for i in range(300)]
    r1, r2 = make_reads(300, 7), make_reads(200, 8)
    f1, f2 = str(tmp_path / "a.fq"), str(tmp_path / "b.fq")
    write_fastq(f1, names, r1); write_fastq(f2, names[:200], r2)
    p = run(null_driver, ["--seq-mode", "2", "--max-reads", "64", f1, f2, d, str(tmp_path), "job"], check=False)
    assert p.returncode == 1 and "mate file is shorter" in p.stderr.decode()
    p = run(null_driver, ["--seq-mode", "1", str(tmp_path / "missing.fq"), d, str(tmp_path), "job"], check=False)
    assert p.returncode == 1 and p.stderr
    p = run(null_driver, ["--seq-mode", "1", f1, str(tmp_path / "no_db"), str(tmp_path), "job"], check=False)
    assert p.returncode == 1 and "no taxonomy" in p.stderr.decode()
    p = run(null_driver, ["--seq-mode", "1", "--devices", "0,9", f1, d, str(tmp_path), "job"], check=False)
    assert p.returncode == 1 and "bad device ordinal" in p.stderr.decode()
    p = run(null_driver, ["--seq-mode", "1", "--mask", "1", f1, d, str(tmp_path), "job"], check=False)
    assert p.returncode == 1 and "not implemented" in p.stderr.decode()
    p = run(null_driver, ["--seq-mode", "1", f1, d, str(tmp_path)], check=False)
    assert p.returncode == 1 and "usage" in p.stderr.decode()
    p = run(null_driver, ["--seq-mode", "1", f1, d, str(tmp_path / "no_such_dir"), "job"], check=False)
    assert p.returncode == 1 and "cannot write" in p.stderr.decode()
    # an empty input: header lines only, an empty report
    f0 = str(tmp_path / "empty.fq"); open(f0, "w").close()
    run(null_driver, ["--seq-mode", "1", f0, d, str(tmp_path), "job0"])
    assert open(tmp_path / "job0_classifications.tsv").read().count("\n") == 1


def test_filter_command_outputs(null_driver, null_db, tmp_path):
    """`metabuli filter` (QueryFilter.cpp:75-118): <base>_filtered.fna holds the reads NOT classified, with --print-mode 2 <base>_removed.fna the others"""
    d, tax, ids = null_db
    n = 500
    names = [f"p{i}" for i in range(n)]
    r1, r2 = make_reads(n, 9, odd=False), make_reads(n, 10, odd=False)
    f1, f2 = str(tmp_path / "sample_1.fq"), str(tmp_path / "sample_2.fq")
    write_fastq(f1, names, r1); write_fastq(f2, names, r2)
    res, _, _ = predict(r1, r2, ids)
    run(null_driver, ["--seq-mode", "2", "--filter", "1", "--print-mode", "2", "--max-reads", "120", "--devices", "0,1", f1, f2, d])
    for path, reads in ((tmp_path / "sample_1", r1), (tmp_path / "sample_2", r2)):
        kept = b"".join(b">" + nm.encode() + b"\n" + s + b"\n" for nm, s, r in zip(names, reads, res) if not r["is_classified"])
        gone = b"".join(b">" + nm.encode() + b"\n" + s + b"\n" for nm, s, r in zip(names, reads, res) if r["is_classified"])
        assert open(str(path) + "_filtered.fna", "rb").read() == kept
        assert open(str(path) + "_removed.fna", "rb").read() == gone
    assert os.path.exists(str(tmp_path / "sample_1") + "_classifications.tsv") and os.path.exists(str(tmp_path / "sample_1") + "_report.tsv")


def test_partitioned_driver_path(null_driver, null_db, tmp_path):
    """--partitioned 1: one call per batch for all engines together (mtb_classify_batch_partitioned), the text of the reads"""
    d, tax, ids = null_db
    n = 400
    names = [f"r{i}" for i in range(n)]
    r1 = make_reads(n, 11)
    fq = str(tmp_path / "r.fq"); write_fastq(fq, names, r1)
    want_c, want_r, _, _, _ = expected_files(str(tmp_path), tax, ids, names, r1, None)
    run(null_driver, ["--seq-mode", "1", "--max-reads", "90", "--partitioned", "1", "--devices", "0,1,2", fq, d, str(tmp_path), "job"])
    assert open(tmp_path / "job_classifications.tsv").read() == want_c
    assert same_report(open(tmp_path / "job_report.tsv").read(), want_r)


def test_krona_file(null_driver, null_db, tmp_path):
    d, tax, ids = null_db
    n = 300
    names = [f"r{i}" for i in range(n)]
    r1 = make_reads(n, 12)
    fq = str(tmp_path / "r.fq"); write_fastq(fq, names, r1)
    _, _, _, counts, tv = expected_files(str(tmp_path), tax, ids, names, r1, None)
    run(null_driver, ["--seq-mode", "1", fq, d, str(tmp_path), "job"])
    krona = open(tmp_path / "job_krona.html").read()
    nodes = rs.krona_nodes(tv, counts, n)
    body = krona[krona.index('<node name="all">'):-len("</krona></div></body></html>")]
    assert sorted(body.split("<node ")) == sorted(nodes.split("<node ")) and body.count("</node>") == nodes.count("</node>")
    assert "Odd &lt;name&gt; &amp; &quot;quotes&quot;" in krona or counts.get(max(tax.parent), 0) == 0
