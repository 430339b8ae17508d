"""Block-parallel inflation of ordinary gzip files (metabuli_amd/csrc/host/pgzip.h: guessed block starts, unknown windows as
marker symbols, exact-meeting rule between chunks, member CRC / length checks) against Python's gzip, and through the FASTA/FASTQ
reader.  No GPU needed."""
import gzip
import hashlib
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def tools(tmp_path_factory):
    d = tmp_path_factory.mktemp("pgz")
    chk, dump = str(d / "pgzip_check"), str(d / "fastx_dump")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-pthread", "-o", chk, os.path.join(ROOT, "tests", "emu", "pgzip_check.cpp"), "-lz"])
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-pthread", "-o", dump, os.path.join(ROOT, "tests", "emu", "fastx_dump.cpp"), "-lz"])
    return chk, dump


def _fastq(n, seed, L=150):
    r = np.random.default_rng(seed)
    bases = r.choice(np.frombuffer(b"ACGT", np.uint8), size=(n, L))
    qual = r.choice(np.frombuffer(b"FFFFFFFF:::,,#", np.uint8), size=(n, L))
    out = []
    for i in range(n):
        out.append(b"@read%d len=%d\n" % (i, L) + bases[i].tobytes() + b"\n+\n" + qual[i].tobytes() + b"\n")
    return b"".join(out)


def _inflate(chk, path, threads, chunk, want=None):
    cmd = [chk, str(path), str(threads), str(chunk)] + ([str(want)] if want else [])
    return subprocess.run(cmd, capture_output=True, timeout=300)


@pytest.fixture(scope="module")
def text():
    return _fastq(20000, 3)                         # 6.4 MB


@pytest.mark.parametrize("level", [1, 6, 9])
def test_equals_gzip_for_every_level_chunking_and_thread_count(tools, tmp_path, text, level):
    p = tmp_path / "t.gz"
    p.write_bytes(gzip.compress(text, compresslevel=level))
    for threads, chunk in ((1, 1 << 20), (4, 65536), (8, 200000), (3, 70001)):
        r = _inflate(tools[0], p, threads, chunk)
        assert r.returncode == 0, r.stderr
        assert r.stdout == text, (level, threads, chunk)
    r = _inflate(tools[0], p, 4, 65536, want=50000)              # many small requests
    assert r.returncode == 0 and r.stdout == text


def test_members_stored_blocks_tiny_and_empty_files(tools, tmp_path, text):
    chk = tools[0]
    p = tmp_path / "multi.gz"                                    # several members, different levels, zero padding behind the last one
    p.write_bytes(gzip.compress(text[:2_000_000], 6) + gzip.compress(text[2_000_000:4_500_000], 1) + gzip.compress(text[4_500_000:], 9) + b"\0" * 100)
    for threads, chunk in ((1, 1 << 20), (4, 65536), (8, 150000)):
        r = _inflate(chk, p, threads, chunk)
        assert r.returncode == 0 and r.stdout == text, (threads, chunk, r.stderr)
    for data, level in ((b"", 6), (b"ACGT\n", 6), (text[:300000], 0), (b"\n".join([b"A" * 70] * 5000), 9)):
        p = tmp_path / "s.gz"
        p.write_bytes(gzip.compress(data, compresslevel=level))
        r = _inflate(chk, p, 4, 65536)
        assert r.returncode == 0 and r.stdout == data, (len(data), level, r.stderr)


def test_damaged_files_are_errors(tools, tmp_path, text):
    chk = tools[0]
    good = gzip.compress(text, 6)
    bad = bytearray(good); bad[len(bad) // 2] ^= 0x55
    p = tmp_path / "bad.gz"; p.write_bytes(bad)
    r = _inflate(chk, p, 4, 65536)
    assert r.returncode != 0 and b"error" in r.stderr and r.stdout != text
    p.write_bytes(good[: len(good) // 3])                         # truncated
    r = _inflate(chk, p, 4, 65536)
    assert r.returncode != 0 and b"error" in r.stderr
    crc = bytearray(good); crc[-6] ^= 1                           # trailer CRC
    p.write_bytes(crc)
    r = _inflate(chk, p, 4, 65536)
    assert r.returncode != 0 and b"CRC" in r.stderr


def test_reader_takes_big_gzip_files_through_the_parallel_inflater(tools, tmp_path, monkeypatch):
    """> 4 MB of ordinary gzip: the FASTQ reader maps the file and inflates it block-parallel; same records as from the plain file and
    as through the zlib stream (MTB_NO_PGZIP)"""
    dump = tools[1]
    text = _fastq(60000, 9)                                       # 19 MB -> ~5.5 MB of gzip
    plain = tmp_path / "r.fq"; plain.write_bytes(text)
    gz = tmp_path / "r.fq.gz"; gz.write_bytes(gzip.compress(text, 4))
    assert gz.stat().st_size > (4 << 20)
    def digest(path, env=None):
        out = subprocess.run([dump, str(path), "4", str(8 << 20), "20000"], capture_output=True, check=True, env=env).stdout
        return hashlib.sha1(out).hexdigest(), out.count(b"\n")
    want = digest(plain)
    assert want[1] == 60000
    assert digest(gz) == want
    assert digest(gz, env=dict(os.environ, MTB_NO_PGZIP="1")) == want


def test_other_compressor_strategies_and_binary_data(tools, tmp_path, text):
    """fixed-Huffman-only, Huffman-only and run-length streams, binary data (nothing passes the finder's text test: one thread inflates it
    all), and a text that only starts after megabytes of binary"""
    import zlib
    chk = tools[0]
    rng = np.random.default_rng(2)
    binary = rng.integers(0, 256, size=3_000_000, dtype=np.uint8).tobytes()
    cases = []
    for strategy in (zlib.Z_FIXED, zlib.Z_HUFFMAN_ONLY, zlib.Z_RLE, zlib.Z_FILTERED):
        co = zlib.compressobj(6, zlib.DEFLATED, 31, 8, strategy)
        cases.append((text[:2_500_000], co.compress(text[:2_500_000]) + co.flush()))
    cases.append((binary, gzip.compress(binary, 6)))
    mixed = binary + text[:3_000_000]
    cases.append((mixed, gzip.compress(mixed, 6)))
    low = b"".join(bytes([65 + (i * 7) % 4]) * (1 + i % 300) for i in range(20000))          # long runs: matches at distance 1, overlapping copies
    cases.append((low, gzip.compress(low, 9)))
    for k, (plain, comp) in enumerate(cases):
        assert gzip.decompress(comp) == plain
        p = tmp_path / f"c{k}.gz"; p.write_bytes(comp)
        for threads, chunk in ((4, 65536), (2, 1 << 20)):
            r = _inflate(chk, p, threads, chunk)
            assert r.returncode == 0 and r.stdout == plain, (k, threads, chunk, r.stderr[:200])


def test_bytes_behind_a_complete_member_end_the_input_like_zlib(tools, tmp_path, text):
    """non-gzip bytes after a valid member: zlib's gzread (the small-file path, the reference's kseq reader) stops there; the
    block-parallel inflater must accept the same file, whatever its size and thread count"""
    chk = tools[0]
    p = tmp_path / "trail.gz"
    p.write_bytes(gzip.compress(text, 6) + b"this is not a gzip header, just bytes somebody appended\n" * 10)
    for threads, chunk in ((1, 1 << 20), (4, 65536), (8, 150000)):
        r = _inflate(chk, p, threads, chunk)
        assert r.returncode == 0 and r.stdout == text, (threads, chunk, r.stderr[:200])
