"""CPU: the threaded host text layer under AddressSanitizer + UBSan and under ThreadSanitizer.

`make -C raft_amd/host asan tsan` builds instrumented copies of libraft_host.so into build/host_san/; this file runs
tests/test_host_io.py against each of them in a child interpreter -- RAFT_HOST_LIB names the library, the sanitizer's
runtime is preloaded because the interpreter itself is not instrumented.  A report makes the child exit non-zero
(ASan aborts; TSan: exitcode=66).  GPU sanitizers are not available on this pool: only the CPU side is covered.
"""
import os
import shutil
import subprocess
import sys

import pytest
from raft_testlib import ROOT

HOST = os.path.join(ROOT, "raft_amd", "host")
SAN = os.path.join(ROOT, "build", "host_san")


def _runtime(name):
    p = subprocess.run(["gcc", f"-print-file-name={name}"], stdout=subprocess.PIPE, text=True).stdout.strip()
    return p if os.path.isabs(p) and os.path.exists(p) else None


def _run(lib, preload, extra_env, select=None):
    env = dict(os.environ, RAFT_HOST_LIB=lib, LD_PRELOAD=":".join(preload), **extra_env)
    cmd = [sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_host_io.py"), "-x", "-q", "-p", "no:cacheprovider"]
    if select:
        cmd += ["-k", select]
    r = subprocess.run(cmd, env=env, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=1500)
    return r.returncode, r.stdout


@pytest.mark.skipif(shutil.which("g++") is None or shutil.which("make") is None, reason="no host toolchain")
def test_host_layer_is_clean_under_asan_and_ubsan():
    rt = [_runtime("libasan.so"), _runtime("libubsan.so")]
    if None in rt:
        pytest.skip("sanitizer runtimes not installed")
    subprocess.run(["make", "-C", HOST, "asan"], check=True, stdout=subprocess.DEVNULL)
    # (leaks are not checked: the interpreter and numpy keep memory until exit)
    rc, out = _run(os.path.join(SAN, "libraft_host_asan.so"), rt, {"ASAN_OPTIONS": "detect_leaks=0:abort_on_error=1", "UBSAN_OPTIONS": "halt_on_error=1"})
    assert rc == 0 and " passed" in out and "ERROR: AddressSanitizer" not in out and "runtime error:" not in out, out[-3000:]


@pytest.mark.skipif(shutil.which("g++") is None or shutil.which("make") is None, reason="no host toolchain")
def test_host_layer_is_clean_under_tsan():
    rt = [_runtime("libtsan.so")]
    if None in rt:
        pytest.skip("sanitizer runtime not installed")
    subprocess.run(["make", "-C", HOST, "tsan"], check=True, stdout=subprocess.DEVNULL)
    # the tests that start child processes are left out: a fork() from the multi-threaded, TSan-preloaded interpreter deadlocks
    # in the sanitizer's own fork handling (they run under ASan above and in the plain suite)
    rc, out = _run(os.path.join(SAN, "libraft_host_tsan.so"), rt, {"TSAN_OPTIONS": "report_signal_unsafe=0:exitcode=66"},
                   select="not split_naive_matches_reference and not fastq")
    assert rc == 0 and " passed" in out and "WARNING: ThreadSanitizer" not in out, out[-3000:]
