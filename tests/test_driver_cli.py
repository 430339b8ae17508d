"""Command line of the stand-alone driver (metabuli_amd/csrc/host/classify_main.cpp): what is parsed before any GPU work.
No GPU needed: flags that would change the answers and are not implemented must be refused, unknown flags rejected, the
reference's no-op flags accepted, and a machine without a GPU must get a loud error, not a result."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "metabuli_amd", "csrc")


@pytest.fixture(scope="module")
def exe():
    if not os.path.exists(os.path.join(CSRC, "libmtb.so")):
        pytest.skip("libmtb.so not built")
    subprocess.check_call(["make", "-C", CSRC, "mtb_classify"], stdout=subprocess.DEVNULL)
    return os.path.join(CSRC, "mtb_classify")


def _run(exe, *args):
    p = subprocess.run([exe, *args], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    return p.returncode, p.stderr


@pytest.mark.parametrize("flag,value,needle", [("--mask", "1", "--mask 1"), ("--reduced-aa", "1", "--reduced-aa 1")])
def test_flags_that_change_the_answers_are_refused(exe, tmp_path, flag, value, needle):
    rc, err = _run(exe, flag, value, "--seq-mode", "1", "r.fq", str(tmp_path), str(tmp_path), "job")
    assert rc == 1 and needle in err and "not implemented" in err


def test_unknown_flag_and_wrong_arity(exe, tmp_path):
    rc, err = _run(exe, "--no-such-flag", "1", "r.fq", str(tmp_path), str(tmp_path), "job")
    assert rc == 1 and "unknown flag --no-such-flag" in err
    rc, err = _run(exe, "--seq-mode", "2", "r1.fq", str(tmp_path), str(tmp_path), "job")          # the mate is missing
    assert rc == 1 and "usage: mtb_classify" in err and "<FASTA/Q> <FASTA/Q>" in err
    rc, err = _run(exe, "--filter", "1", "--seq-mode", "1", "r.fq", str(tmp_path), "extra")     # filter: <reads> <DBDIR> only
    assert rc == 1 and "usage: mtb_classify --filter 1" in err
    rc, err = _run(exe, "--seq-mode")
    assert rc == 1 and "missing value" in err


def test_reference_flags_without_a_meaning_here_are_accepted(exe, tmp_path):
    """`--max-ram 128 --match-per-kmer 4 --mask 0 ...` must not break a script written for `metabuli classify`: they are
    consumed with a one-line note each; the run then fails later (here: no database / no GPU), not in the parser."""
    rc, err = _run(exe, "--max-ram", "128", "--match-per-kmer", "4", "--mask", "0", "--mask-prob", "0.9", "--hamming-margin", "0", "--max-gap", "0",
                   "--validate-input", "0", "--validate-db", "0", "--print-log", "0", "-v", "3", "--threads", "2", "--seq-mode", "1",
                   "r.fq", str(tmp_path / "nodb"), str(tmp_path), "job")
    assert rc == 1
    assert "unknown flag" not in err and "usage:" not in err
    assert err.count("accepted for compatibility") == 9         # everything above except --mask 0 (silently fine) and --threads / --seq-mode
    assert "mtb_classify:" in err.splitlines()[-1]              # the actual failure comes last and names the program


def test_own_flags_are_parsed(exe, tmp_path):
    """--max-reads / --pack-reads / --gpu-workers / --devices / --partitioned are the driver's own flags: they must get through the
    parser (the run then fails at the database, not at the command line)"""
    rc, err = _run(exe, "--max-reads", "100000", "--pack-reads", "0", "--gpu-workers", "2", "--devices", "0", "--partitioned", "0", "--lineage", "1",
                   "--seq-mode", "1", "r.fq", str(tmp_path / "nodb"), str(tmp_path), "job")
    assert rc == 1 and "unknown flag" not in err and "usage:" not in err and "missing value" not in err
