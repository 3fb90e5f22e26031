"""tools/hipemu, the wave64 CPU emulator under tests/test_emu_lds.py, tested on its own: the answers of the cross-lane
operations, the out-of-lockstep execution between them, and the detection of divergent collectives."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def _build(tmp_path):
    exe = str(tmp_path / "hipemu_selftest")
    emu = os.path.join(ROOT, "tools", "hipemu")
    subprocess.check_call(["g++", "-O1", "-g", "-std=c++17", "-I", emu, os.path.join(HERE, "emu_src", "selftest.cpp"), os.path.join(emu, "hipemu.cpp"),
                           "-o", exe, "-ldl"])
    return exe


def test_hipemu_operations_and_lockstep(tmp_path):
    r = subprocess.run([_build(tmp_path), "ok"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=120)
    assert r.returncode == 0, (r.stdout, r.stderr)
    assert b"hipemu selftest ok" in r.stdout


def test_hipemu_catches_divergent_collectives(tmp_path):
    r = subprocess.run([_build(tmp_path), "diverge"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=120)
    assert r.returncode != 0
    assert b"hipemu:" in r.stderr and (b"divergent" in r.stderr or b"deadlock" in r.stderr), r.stderr
