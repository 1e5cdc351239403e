"""Host-only engine components (cache policy, tracer, offload store, priority block reader) under ASan + UBSan, and the
threaded reader again under ThreadSanitizer."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_host_components_under_asan_ubsan(tmp_path):
    cxx = shutil.which("g++") or shutil.which("clang++")
    if cxx is None:
        pytest.skip("no host C++ compiler")
    exe = os.path.join(tmp_path, "host_fuzz")
    cmd = [cxx, "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=all",
           "-pthread", "-I", os.path.join(ROOT, "moe-infinity_amd", "csrc"), os.path.join(ROOT, "tests", "host_fuzz.cpp"), "-o", exe]
    b = subprocess.run(cmd, capture_output=True, text=True)
    if b.returncode != 0 and "sanitize" in (b.stderr or "").lower() and "cannot find" in b.stderr.lower():
        pytest.skip("sanitizer runtime not installed")
    assert b.returncode == 0, b.stderr[-2000:]
    store = os.path.join(tmp_path, "store")
    r = subprocess.run([exe, store], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "host_fuzz ok" in r.stdout, (r.stdout + r.stderr)[-3000:]


def test_priority_reader_under_tsan(tmp_path):
    cxx = shutil.which("g++") or shutil.which("clang++")
    if cxx is None:
        pytest.skip("no host C++ compiler")
    exe = os.path.join(tmp_path, "host_fuzz_tsan")
    cmd = [cxx, "-std=c++17", "-O1", "-g", "-fsanitize=thread", "-pthread",
           "-I", os.path.join(ROOT, "moe-infinity_amd", "csrc"), os.path.join(ROOT, "tests", "host_fuzz.cpp"), "-o", exe]
    b = subprocess.run(cmd, capture_output=True, text=True)
    if b.returncode != 0 and ("tsan" in (b.stderr or "").lower() or "cannot find" in (b.stderr or "").lower()):
        pytest.skip("thread sanitizer runtime not installed")
    assert b.returncode == 0, b.stderr[-2000:]
    r = subprocess.run([exe, os.path.join(tmp_path, "store")], capture_output=True, text=True, timeout=600)
    if r.returncode != 0 and "unexpected memory mapping" in (r.stdout + r.stderr):
        pytest.skip("ThreadSanitizer cannot run in this container (ASLR layout)")
    assert r.returncode == 0 and "host_fuzz ok" in r.stdout and "WARNING: ThreadSanitizer" not in r.stderr, (r.stdout + r.stderr)[-3000:]
