"""Build libmoeinf_hip.so for gfx950 with hipcc (in-tree, no torch dependency).

Every source is compiled to its own object under build/ (git-ignored) and re-compiled only when it or a header is
newer than the object; stale objects are compiled in parallel, then linked."""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(HERE, "build")
LIB = os.path.join(HERE, "libmoeinf_hip.so")
SOURCES = sorted(f for f in os.listdir(CSRC) if f.endswith(".hip")) + ["engine.cpp", "engine_ep.cpp", "capi_host.cpp"]
HEADERS = sorted(f for f in os.listdir(CSRC) if f.endswith(".h")) + [os.path.join("..", "..", "include", "moeinf.h")]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-value", "-Wno-unused-result"]


def _hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    return "hipcc"


def _mtime(p):
    return os.path.getmtime(p) if os.path.exists(p) else 0.0


def _newest_header():
    return max(_mtime(os.path.join(CSRC, h)) for h in HEADERS)


def needs_build():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, s) for s in SOURCES + HEADERS]
    return any(os.path.getmtime(d) > t for d in deps if os.path.exists(d))


def _obj(src):
    return os.path.join(OBJ, os.path.splitext(src)[0] + ".o")


def build(force=False, verbose=False):
    if not force and not needs_build():
        return LIB
    os.makedirs(OBJ, exist_ok=True)
    hdr = _newest_header()
    stale = [s for s in SOURCES if force or _mtime(_obj(s)) < max(_mtime(os.path.join(CSRC, s)), hdr)]

    def compile_one(src):
        if src.endswith(".cpp"):
            # host code only (HIP runtime API, no kernels): a plain C++ compile — hipcc would run the device pass over it too
            cmd = [_hipcc(), "-x", "c++", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include", "-O2", "-std=c++17", "-fPIC",
                   "-Wno-unused-value", "-Wno-unused-result", "-c", os.path.join(CSRC, src), "-o", _obj(src)]
        else:
            cmd = [_hipcc()] + FLAGS + ["-c", os.path.join(CSRC, src), "-o", _obj(src)]
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        subprocess.run(cmd, check=True, cwd=CSRC)

    with ThreadPoolExecutor(max_workers=max(1, min(len(stale), os.cpu_count() or 1))) as ex:
        list(ex.map(compile_one, stale))
    cmd = [_hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + [_obj(s) for s in SOURCES]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.run(cmd, check=True, cwd=CSRC)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
