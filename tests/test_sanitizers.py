"""Sanitizer runs of the host code, in the CPU build container only (GPU AddressSanitizer / XNACK are not
available on the pool and nothing here touches a device).  The reference is single-threaded under the GIL with file-scope
globals (cscore.c:26-34); this library runs three host threads per device with queues, pools and per-thread error strings.

  * oracle/cscore_oracle.c under ASan + UBSan (`make -C oracle asan`), driven by the whole of tests/test_oracle_golden.py;
  * motifscan_amd/csrc/ms_pipeline.h -- the header ms_stream.hip's threads, queues and one-scan-ahead loop are built from --
    under TSan with stub stage functions (tests/sanitize/stream_tsan.cpp);
  * motifscan_amd/csrc/ms_plan.cpp -- thresholds, quantiser, paired rows, row-tile DP, operand image -- under ASan + UBSan
    (tests/sanitize/plan_asan.cpp), on random / degenerate motif sets and on the benchmark set at every cutoff column;
  * motifscan_amd/csrc/ms_hostpack.cpp + ms_numa.cpp -- the host packer and the NUMA look-ups -- under ASan + UBSan (tests/sanitize/host_asan.cpp).
"""
import os
import shutil
import struct
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "motifscan_amd", "csrc")
SAN = os.path.join(ROOT, "tests", "sanitize")

pytestmark = pytest.mark.skipif(shutil.which("g++") is None or shutil.which("make") is None, reason="needs g++ and make")


def _runtime(name):
    out = subprocess.run(["gcc", f"-print-file-name={name}"], capture_output=True, text=True).stdout.strip()
    return out if os.path.isabs(out) and os.path.exists(out) else None


@pytest.fixture(scope="module")
def sanitize_binaries():
    if _runtime("libtsan.so") is None or _runtime("libasan.so") is None:
        pytest.skip("the sanitizer runtimes are not installed on this box")
    subprocess.run(["make", "-s", "-C", CSRC, "sanitize"], check=True)
    return os.path.join(SAN, "stream_tsan.bin"), os.path.join(SAN, "plan_asan.bin")


def test_stream_pipeline_under_thread_sanitizer(sanitize_binaries):
    env = dict(os.environ, TSAN_OPTIONS="halt_on_error=1 second_deadlock_stack=1 exitcode=66")
    for _ in range(3):                                            # thread schedules differ from run to run
        out = subprocess.run([sanitize_binaries[0]], env=env, capture_output=True, text=True, timeout=600)
        assert out.returncode == 0 and "stream_tsan: ok" in out.stdout and "ThreadSanitizer" not in out.stderr, out.stderr[-4000:]


def test_prefilter_planner_under_address_and_ub_sanitizers(sanitize_binaries, tmp_path, jaspar579):
    keys = ["1e-2", "1e-3", "1e-4", "1e-5"]
    widths = np.asarray(jaspar579["widths"], dtype=np.int32)
    cuts = np.stack([jaspar579["cutoffs"][k] for k in keys], axis=1).astype(np.float64)          # [n][n_sets]
    vals = np.asarray(jaspar579["pwm_values"], dtype=np.float64)[:4 * int(widths.sum())]
    path = tmp_path / "motifs.bin"
    with open(path, "wb") as fh:
        fh.write(struct.pack("<ii", len(widths), len(keys)))
        fh.write(widths.tobytes()); fh.write(np.ascontiguousarray(cuts).tobytes()); fh.write(vals.tobytes())
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1")
    out = subprocess.run([sanitize_binaries[1], str(path)], env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0 and "plan_asan: ok" in out.stdout, (out.stdout + out.stderr)[-4000:]
    assert "runtime error" not in out.stderr and "AddressSanitizer" not in out.stderr, out.stderr[-4000:]
    assert int(out.stdout.split("(")[1].split()[0]) > 300


def test_host_packer_and_numa_lookups_under_address_and_ub_sanitizers(sanitize_binaries):
    """Round 6's host-only sources from the library's own files (ms_hostpack.cpp: the packer behind the genome file and MS_STREAM_HOST_PACK,
    AVX2 and scalar paths on exact-size buffers; ms_numa.cpp: the cpulist parser and sysfs reads on malformed and missing inputs)."""
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1")
    out = subprocess.run([os.path.join(SAN, "host_asan.bin")], env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "host_asan: ok" in out.stdout, (out.stdout + out.stderr)[-4000:]
    assert "runtime error" not in out.stderr and "AddressSanitizer" not in out.stderr, out.stderr[-4000:]


def test_oracle_restatement_under_address_and_ub_sanitizers():
    asan = _runtime("libasan.so")
    if asan is None:
        pytest.skip("libasan is not installed on this box")
    subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "asan"], check=True)
    so = os.path.join(ROOT, "oracle", "liboracle_asan.so")
    # the interpreter itself is not instrumented: leak reports would be CPython's own; everything else stops the run
    env = dict(os.environ, LD_PRELOAD=asan, ORACLE_SO=so, ASAN_OPTIONS="detect_leaks=0:abort_on_error=1", UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1")
    out = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_oracle_golden.py"), "-x", "-q", "-p", "no:cacheprovider"],
                         env=env, capture_output=True, text=True, timeout=1800, cwd=ROOT)
    assert out.returncode == 0 and " passed" in out.stdout, (out.stdout + out.stderr)[-4000:]
    assert "runtime error" not in out.stderr and "AddressSanitizer" not in out.stderr, out.stderr[-4000:]
