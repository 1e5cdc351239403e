"""The oracle is test infrastructure: nothing under the product package (Python or C++/HIP) may import, include, link or
execute anything under oracle/ — only tests/, __graft_entry__.smoke() and bench.py's checker legs do."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "moe-infinity_amd")


def _files(top, exts):
    for d, _, names in os.walk(top):
        if os.path.basename(d) in ("build", "__pycache__"):
            continue
        for n in names:
            if n.endswith(exts):
                yield os.path.join(d, n)


def test_no_python_module_of_the_package_imports_the_oracle():
    pat = re.compile(r"^\s*(from\s+oracle\b|import\s+oracle\b|from\s+\.+\s*oracle\b)|importlib\.import_module\(\s*['\"]oracle", re.M)
    bad = [p for p in _files(PKG, (".py",)) if pat.search(open(p).read())]
    bad += [p for p in _files(os.path.join(ROOT, "moe_infinity_amd"), (".py",)) if pat.search(open(p).read())]
    assert not bad, bad


def test_no_native_source_includes_or_links_anything_under_oracle():
    for p in list(_files(os.path.join(PKG, "csrc"), (".cpp", ".hip", ".h"))) + [os.path.join(PKG, "build.py"), os.path.join(ROOT, "include", "moeinf.h")]:
        src = open(p).read()
        assert not re.search(r'#\s*include\s*[<"][^>"]*oracle', src), p
        assert "oracle/_ref" not in src and "libmoeinf_ref" not in src, p


def test_bench_uses_the_oracle_only_in_its_checker_legs():
    src = open(os.path.join(ROOT, "bench.py")).read()
    lines = [i for i, l in enumerate(src.splitlines()) if re.match(r"\s*from oracle\b|\s*import oracle\b", l)]
    assert lines, "bench.py's cpu_baseline / parity legs are expected to use the oracle"
    body = src.splitlines()
    for i in lines:
        if "synth" in body[i]:  # seeded synthetic inputs (activations, weights): data generation, shared with the tests
            continue
        ctx = "\n".join(body[max(0, i - 12): i + 1])
        assert "no_cpu_baseline" in ctx or "cpu_baseline" in ctx or "parity" in ctx.lower(), f"bench.py:{i + 1} imports the oracle outside the checker legs"
