"""CPU: the C-ABI library loads without a GPU and exports every symbol include/raft_hip.h declares."""
import os
import re

import pytest
from raft_testlib import ROOT

from raft_amd import engine
from raft_amd.params import RaftParams


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "raft_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(raft_hip_[a-z0-9_]+)\s*\(", text)))


def test_header_and_binding_agree():
    assert declared_symbols() == sorted(engine.EXPORTS)


def test_library_exports_every_symbol():
    lib = engine.load_library()
    for name in declared_symbols():
        assert hasattr(lib, name), name
    assert lib.raft_hip_abi_version() == 11
    assert lib.raft_hip_strerror(2).decode().startswith("PAF record names a read id")


def test_cli_binary_fails_loudly_without_device(tmp_path):
    import subprocess
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    exe = os.path.join(ROOT, "raft_amd", "bin", "raft")
    (tmp_path / "a.fa").write_text(">x\nACGT\n")
    (tmp_path / "b.paf").write_text("x\t4\t0\t4\t+\tx\t4\t0\t4\t1\t1\t1\n")
    r = subprocess.run([exe, "-e", "3", "a.fa", "b.paf"], cwd=tmp_path, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    assert r.returncode == 1 and b"ERROR, raft_hip_create(), device 0: HIP device/runtime error" in r.stdout


def test_no_cpu_fallback_without_device():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(engine.RaftError) as e:
        engine.Engine(RaftParams(est_cov=30))
    assert e.value.code == engine.ERR_DEVICE


def test_invalid_params_rejected_before_any_device_work():
    lib = engine.load_library()
    import ctypes as C
    ctx = C.c_void_p()
    bad = engine._cparams(RaftParams(est_cov=0))
    assert lib.raft_hip_create(0, C.byref(bad), C.byref(ctx)) == engine.ERR_PARAM
    bad = engine._cparams(RaftParams(est_cov=3, read_length=100, interval_length=200, repeat_length=200))
    assert lib.raft_hip_create(0, C.byref(bad), C.byref(ctx)) == engine.ERR_PARAM


def test_library_exports_nothing_but_the_abi():
    """Built with -fvisibility=hidden and a version script: kernel handles, template instances of the standard library and the
    engine's internals stay local (`nm -D | grep -v raft_hip_` empty)."""
    import shutil
    import subprocess
    if shutil.which("nm") is None:
        pytest.skip("no nm")
    out = subprocess.run(["nm", "-D", "--defined-only", engine._LIB_PATH], stdout=subprocess.PIPE, text=True, check=True).stdout
    names = [ln.split()[-1] for ln in out.splitlines() if ln.strip()]
    assert names and all(n.startswith("raft_hip_") for n in names), [n for n in names if not n.startswith("raft_hip_")][:10]
    assert sorted(names) == declared_symbols()
