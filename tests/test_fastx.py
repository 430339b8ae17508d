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


def _run(dump, path, threads, block, batch, *extra):
    out = subprocess.run([dump, path, str(threads), str(block), str(batch)] + list(extra), capture_output=True, check=True).stdout.decode()
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


def _bgzf(data: bytes, block=5000) -> bytes:
    """blocked gzip as bgzip writes it (SAM specification 4.1): independent deflate blocks with their size in the 'BC' extra field"""
    import struct
    import zlib
    out = bytearray()
    for o in list(range(0, len(data), block)) + [None]:
        chunk = b"" if o is None else data[o:o + block]           # the last, empty block is the end-of-file marker
        c = zlib.compressobj(6, zlib.DEFLATED, -15)
        comp = c.compress(chunk) + c.flush()
        bsize = 12 + 6 + len(comp) + 8
        out += struct.pack("<BBBBIBBH", 0x1F, 0x8B, 8, 4, 0, 0, 0xFF, 6) + b"BC" + struct.pack("<HH", 2, bsize - 1)
        out += comp + struct.pack("<II", zlib.crc32(chunk) & 0xFFFFFFFF, len(chunk))
    return bytes(out)


def test_bgzf_is_inflated_block_parallel_and_two_bit_packing(dump, tmp_path):
    rng = np.random.default_rng(99)
    recs, text = _records(rng, 900, True)
    pz = tmp_path / "b.fq.gz"
    pz.write_bytes(_bgzf(text.encode()))
    assert gzip.decompress(pz.read_bytes()).decode() == text                # a valid multi-member gzip file as well
    for threads, block, batch in ((4, 3000, 128), (8, 1 << 20, 10**6), (2, 64, 33)):
        assert _run(dump, str(pz), threads, block, batch) == recs, (threads, block, batch)
    # 2-bit codes + invalid mask (what crosses PCIe): the extractor's base classes -- A R W -> A, C M S -> C,
    # H T Y -> T, B D G K U -> G, everything else invalid
    cls = {}
    for k, v in (("ARWarw", "A"), ("CMScms", "C"), ("HTYhty", "T"), ("BDGKUbdgku", "G")):
        for ch in k:
            cls[ch] = v
    exp = [(n, "".join(cls.get(ch, "N") for ch in s)) for n, s in recs]
    p = tmp_path / "x.fq"
    p.write_bytes(text.encode())
    assert _run(dump, str(p), 4, 3000, 100, "pack") == exp
    assert _run(dump, str(pz), 3, 2000, 77, "pack") == exp


def test_damaged_bgzf_blocks_are_errors(dump, tmp_path):
    """block CRC32 is checked, ISIZE beyond the format's 64 KiB is refused, a truncated 'BC' subfield is not read past the mapping"""
    rng = np.random.default_rng(5)
    recs, text = _records(rng, 400, True)
    good = bytearray(_bgzf(text.encode()))
    # flip one bit of the first block's CRC32 (8 bytes before the end of the block; BSIZE at offset 16)
    bsize = (good[16] | (good[17] << 8)) + 1
    bad = bytearray(good); bad[bsize - 8] ^= 1
    p = tmp_path / "crc.fq.gz"; p.write_bytes(bad)
    r = subprocess.run([dump, str(p), "4", "3000", "128"], capture_output=True)
    assert r.returncode != 0 and b"CRC" in r.stderr
    bad = bytearray(good); bad[bsize - 2] = 0x7F                  # ISIZE claims gigabytes
    p.write_bytes(bad)
    r = subprocess.run([dump, str(p), "4", "3000", "128"], capture_output=True)
    assert r.returncode != 0 and b"BGZF" in r.stderr
    p.write_bytes(good[:17])                                     # the mapping ends inside the BC subfield
    r = subprocess.run([dump, str(p), "4", "3000", "128"], capture_output=True)
    assert r.returncode != 0


def test_plain_gzip_stream_longer_than_the_inflaters_queue(dump, tmp_path):
    """A plain (non-BGZF) gzip file is inflated by a background thread 16 MB at a time through a bounded queue: 100 MB of FASTQ
    (more than the queue holds, many refills) must parse exactly like the uncompressed file, and a reader that is destroyed with
    the stream half read must shut its thread down (the truncated-file case ends with an error, not a hang)."""
    import hashlib
    rng = np.random.default_rng(5)
    seq = "".join(rng.choice(list("ACGT"), size=150))
    one = "".join(f"@read{i}\n{seq}\n+\n{'I' * 150}\n" for i in range(2000))
    text = (one * 160).encode()                                  # ~100 MB, 320 000 records
    p = tmp_path / "big.fq"; p.write_bytes(text)
    pz = tmp_path / "big.fq.gz"
    with gzip.open(pz, "wb", compresslevel=1) as f:
        f.write(text)
    def digest(path):
        out = subprocess.run([dump, str(path), "4", str(64 << 20), "50000"], capture_output=True, check=True).stdout
        return hashlib.sha1(out).hexdigest(), out.count(b"\n")
    a, b = digest(p), digest(pz)
    assert a == b and a[1] == 320000
    # truncated stream: an error, not a hang
    pt = tmp_path / "cut.fq.gz"
    pt.write_bytes(pz.read_bytes()[: pz.stat().st_size // 2])
    r = subprocess.run([dump, str(pt), "4", str(64 << 20), "50000"], capture_output=True, timeout=120)
    assert r.returncode != 0 and b"error" in r.stderr


def test_reads_from_a_pipe(dump, tmp_path):
    """a FIFO (process substitution) cannot be mapped or looked ahead in: it goes through the zlib stream reader, plain or gzip"""
    rng = np.random.default_rng(3)
    recs, text = _records(rng, 500, True)
    for payload in (text.encode(), gzip.compress(text.encode())):
        fifo = tmp_path / "in.fifo"
        if fifo.exists():
            fifo.unlink()
        os.mkfifo(fifo)
        import threading
        def feed():
            with open(fifo, "wb") as f:
                f.write(payload)
        t = threading.Thread(target=feed); t.start()
        got = _run(dump, str(fifo), 4, 1 << 20, 100)
        t.join()
        assert got == recs


def test_malformed_fastq_is_refused(dump, tmp_path):
    """wrapped sequence lines, a missing '+' line or a quality string of another length: an error that names the record, not a guess"""
    good = "@r1\nACGTACGT\n+\nIIIIIIII\n@r2\nACGT\n+\nIIII\n"
    for bad in ("@r1\nACGT\nACGT\n+\nIIII\nIIII\n",            # wrapped
                "@r1\nACGTACGT\nIIIIIIII\n@r2\nACGT\n+\nIIII\n",  # no '+' line
                "@r1\nACGTACGT\n+\nIII\n@r2\nACGT\n+\nIIII\n"):   # quality too short
        p = tmp_path / "bad.fq"
        p.write_text(bad)
        r = subprocess.run([dump, str(p), "2", "1000", "10"], capture_output=True)
        assert r.returncode != 0 and b"malformed FASTQ" in r.stderr, (bad, r.stderr)
    p = tmp_path / "good.fq"; p.write_text(good)
    assert _run(dump, str(p), 2, 1000, 10) == [("r1", "ACGTACGT"), ("r2", "ACGT")]
