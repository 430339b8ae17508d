"""The kernels' LOGIC on a machine without a GPU: the library's sources (metabuli_amd/csrc/mtb_api.hip and its kernel headers, unchanged) are
compiled with g++ against tests/hipemu/hip/hip_runtime.h -- a stand-in for the HIP runtime that runs every workgroup as a set of cooperative
fibers, with wave64 ballots / shuffles / DPP / readlane, LDS, barriers and atomics modelled -- and a subset of the GPU parity tests
(tests/test_gpu_parity.py, `-m gpu`) is run against that build in a subprocess (MTB_HIPEMU=1, MTB_LIB=<the emulated build>).

TEST INFRASTRUCTURE ONLY.  It is not a product path (nothing under metabuli_amd/ knows about it; libmtb.so has no CPU path), not a
performance model and not a memory-model checker; the parity claims rest on the `-m gpu` run on an MI355X.  What it buys: kernel logic is
checked against the oracle before GPU minutes are spent, fresh "device" memory is poisoned, and the same build under AddressSanitizer finds
out-of-bounds accesses that a GPU hides (python tests/hipemu/build_emulated.py DIR asan).

The whole `-m gpu` suite through the emulator (about 30 minutes on 8 cores; 221 passed, 0 failed at the end of round 4):
    python tests/hipemu/build_emulated.py /tmp/mtb_hipemu
    MTB_HIPEMU=1 MTB_LIB=/tmp/mtb_hipemu/libmtb_hipemu.so python -m pytest tests -m gpu -q
"""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def emulated_lib(tmp_path_factory):
    sys.path.insert(0, os.path.join(ROOT, "tests", "hipemu"))
    import build_emulated
    d = os.environ.get("MTB_HIPEMU_DIR") or str(tmp_path_factory.mktemp("hipemu"))       # MTB_HIPEMU_DIR: reuse a build between runs
    return build_emulated.build(d)


def _run(lib, select, timeout=1500):
    env = dict(os.environ, MTB_HIPEMU="1", MTB_LIB=lib)
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_gpu_parity.py"), "-m", "gpu", "-q", "-x", "-p", "no:cacheprovider", "-k", select],
                       env=env, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=timeout)
    tail = r.stdout[-3000:]
    assert r.returncode == 0, tail
    assert " passed" in tail and " failed" not in tail, tail
    return tail


def test_emulator_primitives(tmp_path):
    """the wave64 operations of the stand-in runtime against their definitions, on a kernel of its own"""
    src = tmp_path / "prim.cpp"
    src.write_text(r'''
#include <hip/hip_runtime.h>
__global__ void k(uint64_t *out) {
    const uint32_t t = threadIdx.x, lane = t & 63u;
    __shared__ uint32_t s[256];
    s[t] = t * 3u;
    __syncthreads();
    uint64_t acc = s[(t + 1) % blockDim.x];
    acc += __popcll(__ballot(lane % 3 == 0));                                  /* 22 lanes */
    acc += (uint64_t)__shfl((int)lane, 5) + (uint64_t)__shfl_xor((int)lane, 1) + (uint64_t)__shfl_up((int)lane, 2) + (uint64_t)__shfl_down((int)lane, 3);
    acc += (uint64_t)__shfl((int)lane, 3, 16);                                  /* lane 3 of the lane's group of 16 */
    acc += __any(lane == 63) ? 1000 : 0;
    acc += __all(lane < 64) ? 10000 : 0;
    acc += (uint32_t)__builtin_amdgcn_readlane((int)(lane * 7), 9);
    int v = (int)lane;                                                          /* inclusive scan with DPP row shifts / broadcasts (dev_util.h's sequence) */
    v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xF, 0xF, true); v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xF, 0xF, true);
    v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xF, 0xF, true); v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xF, 0xF, true);
    v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xA, 0xF, false); v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xC, 0xF, false);
    acc += (uint64_t)v << 20;
    if (t >= 200) return;                                                        /* ended lanes leave the later operations' active sets */
    acc += (uint64_t)__popcll(__ballot(1)) << 40;
    const int any_or = __syncthreads_or(t == 7);
    out[(uint64_t)blockIdx.x * 256 + t] = acc + ((uint64_t)any_or << 50);
    atomicAdd(&out[65536], (uint64_t)1);
}
int main() {
    uint64_t *d; hipMalloc(&d, 65537 * 8); hipMemset(d, 0, 65537 * 8);
    hipLaunchKernelGGL(k, dim3(256), dim3(256), 0, 0, d);
    unsigned long long bad = 0;
    for (uint32_t b = 0; b < 256; b++) for (uint32_t t = 0; t < 200; t++) {
        const uint32_t lane = t & 63u, w = t >> 6;
        uint64_t e = ((t + 1) % 256) * 3u + 22;
        e += 5 + (lane ^ 1) + (lane >= 2 ? lane - 2 : lane) + (lane + 3 <= 63 ? lane + 3 : lane);
        e += (lane & ~15u) | 3u;
        e += 1000 + 10000 + 9 * 7;
        e += (uint64_t)(lane * (lane + 1) / 2) << 20;
        e += (uint64_t)(w < 3 ? 64 : 8) << 40;                                  /* wavefront 3 keeps lanes 192..199 */
        e += (uint64_t)1 << 50;
        if (d[(uint64_t)b * 256 + t] != e) bad++;
    }
    printf("bad %llu count %llu\n", bad, (unsigned long long)d[65536]);
    return bad != 0 || d[65536] != 256 * 200;
}
''')
    exe = tmp_path / "prim"
    here = os.path.join(ROOT, "tests", "hipemu")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-pthread", "-fno-extern-tls-init", "-I", here, "-include", os.path.join(here, "hipemu_dyn_shared.h"), "-Wno-attributes", "-o", str(exe), str(src)])
    r = subprocess.run([str(exe)], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr


def test_stages_and_fused_batch_on_the_emulator(emulated_lib):
    """every stage entry point and the fused batch (directory join, slot scorers, deferred reads) against the oracle: single-end syncmer toy,
    dense paired-end toy (the large-segment sort with dynamic LDS), legacy format"""
    _run(emulated_lib, "(test_extract_matches_oracle or test_sort_kmers or test_index_decode or test_match_and_sort_matches or test_score or test_fused_batch "
                       "or test_two_bit_reads_give_the_results_of_the_text) and (sync_se or dense_pe or old_format_pe) and not sync_se_acc2")


def test_long_reads_and_long_runs_on_the_emulator(emulated_lib):
    """long reads on ordinal slots (k_join_dir<.., LONG>, k_seg_order, k_score_long) and the wave-cooperative scan of long candidate runs; window tiles
    reading ahead around runs of a dozen candidates"""
    _run(emulated_lib, "(test_fused_batch and sync_long) or (test_long_candidate_runs_are_scanned_by_the_wave and True-1) or test_empty_and_ragged_inputs "
                       "or (test_runs_of_a_dozen_candidates_inside_and_outside_a_window and 11)")


def test_bench_line_of_two_ranks_on_the_emulator(emulated_lib, tmp_path):
    """bench.py's main() launched the way the driver launches bench.py for N = 2 (torch.distributed.run, one rank per GPU; here gloo and the
    emulated build, through tests/hipemu/bench_emulated.py, which hands main() a CPU device): every rank classifies its own reads against its own replica, the time is the maximum over the ranks, rank 0 prints ONE
    JSON line for the whole job"""
    import json
    env = dict(os.environ, MTB_HIPEMU="1", MTB_LIB=emulated_lib, HIPEMU_THREADS="2")
    port = 29600 + os.getpid() % 300
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(port),
                        os.path.join(ROOT, "tests", "hipemu", "bench_emulated.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--reads", "3000", "--targets", "1.5e6", "--species", "8",
                        "--genome-len", "80000", "--filler-species", "2000", "--dist-backend", "gloo", "--shared-gpu"],
                       env=env, cwd=str(tmp_path), capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.strip().split("\n") if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["steps"] == 2 and line["warmup"] == 1 and line["scaling"] == "weak" and line["unit"] == "Mreads/s"
    assert abs(line["value"] - 2 * 3000 * 2 / (line["ms_per_step"] * 2 / 1e3) / 1e6) < 1e-6 * max(1.0, line["value"])      # whole-job reads / max-over-ranks time
    assert line["config"]["classified_fraction"] > 0.5 and {"bound", "achieved", "peak", "frac", "traffic"} <= set(line["roofline"])      # (which kernel dominates is the executor's business)


def test_bench_line_with_every_leg_stays_under_8_kb(emulated_lib, tmp_path):
    """the driver could not parse round 5's 20.6 KB line.  bench.py with EVERY leg on (pairs, long reads, 24-genome reads,
    held-out genomes; parity samples and the CPU baseline too) on the emulated build: ONE JSON line on stdout, shorter than 8 KB, with the
    contract's keys and the short forms of roofline / cpu_baseline / parity_sample / other_configs; the histograms, footprints, notes and
    per-kernel tables are in bench_detail.json (working directory), named by the line; no `frac` above 1 anywhere."""
    import json
    env = dict(os.environ, MTB_HIPEMU="1", MTB_LIB=emulated_lib, HIPEMU_THREADS="4")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "hipemu", "bench_emulated.py"), "--steps", "1", "--warmup", "1", "--reads", "3000", "--targets", "3e6",
                        "--species", "200", "--genome-len", "12000", "--filler-species", "2000", "--leg-pairs", "1000", "--leg-long", "40", "--leg-long-len", "3000",
                        "--leg-novel", "1000", "--heldout", "20", "--cpu-reads", "1000", "--full-parity-reads", "256", "--long-parity-reads", "20"],
                       env=env, cwd=str(tmp_path), capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    out_lines = [ln for ln in r.stdout.split("\n") if ln.strip()]
    assert len(out_lines) == 1, r.stdout[-2000:]
    assert len(out_lines[0]) < 8192, len(out_lines[0])
    line = json.loads(out_lines[0])
    assert {"metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config",
            "roofline", "cpu_baseline", "parity_sample", "other_configs", "stage_ms", "detail"} <= set(line)
    assert {"bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "avg_launch_ms", "launches", "algorithmic_bytes_per_launch"} <= set(line["roofline"])
    assert {"value", "unit", "cores", "kind", "sample"} <= set(line["cpu_baseline"]) and line["cpu_baseline"]["kind"] == "port"
    assert line["parity_sample"]["mismatches"] == 0 and line["parity_sample"]["matches"] == line["parity_sample"]["oracle_matches"]
    assert set(line["other_configs"]) == {"paired", "long", "best_case", "novel"}
    for name, leg in line["other_configs"].items():
        assert "frac" not in leg and leg["ms_per_step"] > 0
        if name != "best_case":
            assert leg["parity"]["mismatches"] == 0
    detail = json.load(open(tmp_path / line["detail"]))
    assert detail["value"] == pytest.approx(line["value"], rel=1e-5) and "roofline_all" in detail and "run_lengths" in detail

    def fracs(x):
        if isinstance(x, dict):
            for k, v in x.items():
                if k.startswith("frac") and isinstance(v, (int, float)):
                    yield v
                yield from fracs(v)
        elif isinstance(x, list):
            for v in x:
                yield from fracs(v)
    assert all(f <= 1.0 for f in fracs(detail)), [f for f in fracs(detail) if f > 1.0]
