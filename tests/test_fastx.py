"""Host ingest (metabuli_amd/csrc/host/fastx.h): the block-parallel FASTA/FASTQ/gzip parser against a line-by-line
Python parser, with block sizes small enough that every record straddles a block or thread boundary."""
import gzip
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def dump(tmp_path_factory):
    exe = str(tmp_path_factory.mktemp("fx") / "fastx_dump")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-pthread", "-o", exe, os.path.join(ROOT, "tests", "emu", "fastx_dump.cpp"), "-lz"])
    return exe


def _run(dump, path, threads, block, batch):
    out = subprocess.run([dump, path, str(threads), str(block), str(batch)], capture_output=True, check=True).stdout.decode()
    return [tuple(l.split("\t")) for l in out.split("\n") if l]


def _records(rng, n, fastq, crlf=False):
    recs, text = [], []
    nl = "\r\n" if crlf else "\n"
    for i in range(n):
        L = int(rng.integers(0, 300)) if i % 17 else 0
        seq = "".join(rng.choice(list("ACGTNacgtRY"), size=L))
        name = f"r{i}_{int(rng.integers(0, 10**6))}"
        recs.append((name, seq))
        if fastq:
            # quality strings that start with '@' or '+' are the hard case for finding record starts
            q = "".join(rng.choice(list("@+IJ#5"), size=L))
            text.append(f"@{name} some comment{nl}{seq}{nl}+{nl}{q}{nl}")
        else:
            w = int(rng.integers(20, 80))
            body = nl.join(seq[k:k + w] for k in range(0, L, w))
            text.append(f">{name}\tdesc{nl}{body}{nl}" if L else f">{name}{nl}")
    return recs, "".join(text)


@pytest.mark.parametrize("fastq", [True, False])
@pytest.mark.parametrize("crlf", [False, True])
def test_parser_matches_reference_parse(dump, tmp_path, fastq, crlf):
    rng = np.random.default_rng(7 + fastq + 2 * crlf)
    recs, text = _records(rng, 700, fastq, crlf)
    p = tmp_path / "x.txt"
    p.write_bytes(text.encode())
    for threads, block, batch in ((1, 1 << 20, 10**6), (4, 700, 50), (3, 4096, 333), (8, 64, 7)):
        assert _run(dump, str(p), threads, block, batch) == recs, (threads, block, batch)
    # no newline at the end of the file, and a gzip copy
    p2 = tmp_path / "y.txt"
    p2.write_bytes(text.rstrip("\r\n").encode())
    assert _run(dump, str(p2), 4, 1000, 100) == recs
    pz = tmp_path / "z.gz"
    with gzip.open(pz, "wb") as f:
        f.write(text.encode())
    assert _run(dump, str(pz), 4, 5000, 128) == recs


def test_empty_and_garbage_inputs(dump, tmp_path):
    p = tmp_path / "e.fq"
    p.write_bytes(b"")
    assert _run(dump, str(p), 2, 1000, 10) == []
    p.write_bytes(b"\n\n")
    assert _run(dump, str(p), 2, 1000, 10) == []
    p.write_bytes(b"hello\n")
    r = subprocess.run([dump, str(p), "2", "1000", "10"], capture_output=True)
    assert r.returncode != 0 and b"neither FASTA nor FASTQ" in r.stderr
