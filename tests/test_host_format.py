"""Host driver's number formatting (metabuli_amd/csrc/host/format.h) against printf: the score column of the classification rows is an
`ostream << float` in the reference (Reporter.cpp:50) = printf("%g"); the driver prints it with integer arithmetic.  No GPU needed."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_float_and_integer_formatting_equal_printf(tmp_path):
    exe = str(tmp_path / "format_check")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-o", exe, os.path.join(ROOT, "tests", "emu", "format_check.cpp")])
    # stride 61 over the bit patterns of [1e-5, 2e6] (5 M floats) + every float of [0.999, 1] + the neighbours of the powers of ten
    out = subprocess.check_output([exe, "61"], text=True)
    assert out.startswith("OK "), out
    assert int(out.split()[1]) > 5_000_000


def test_batched_path_combination_equals_the_serial_loop(tmp_path):
    """tests/emu/combine_batched_check.cpp: the greedy combination of a species' paths evaluated 64 (and 7) candidates at a time =
    the reference's one-at-a-time loop (accepted paths, trims, score bits) on random path sets -- the wave-parallel form the long-read
    scorer's combination phase is to take next"""
    exe = str(tmp_path / "combine_check")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-o", exe, os.path.join(ROOT, "tests", "emu", "combine_batched_check.cpp")])
    out = subprocess.check_output([exe, "5000"], text=True)
    assert out.startswith("OK "), out
