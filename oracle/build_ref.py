"""Build oracle/_ref/libmoeinf_ref.so: two host-only sources of the REFERENCE, compiled where they lie under
/root/reference, plus oracle/ref_build/ref_driver.cpp (C entry points).  TEST INFRASTRUCTURE ONLY.

    core/parallel/expert_module.cpp    expert FFN modules (R6) — torch::matmul/silu/relu/mul on CPU tensors
    core/aio/archer_tensor_index.cpp   archer_index (de)serializer (disk-tier format)

Everything else of the reference's C++ core is unbuildable here (CUDA runtime calls, c10/cuda allocator, its own cmake /
op_builder) — DESIGN.md section 6.  The two files above only PARSE a few CUDA names through the headers they include;
oracle/ref_build/cuda_names/ supplies those names (declarations only, nothing is called).  Needs libtorch (the
PyTorch of this image).  No reference source is copied into the repo; the output is git-ignored and travels to the
GPU box with the snapshot (where /root/reference does not exist and this script is a no-op)."""
import os
import subprocess
import sys
import sysconfig

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"
OUT_DIR = os.path.join(HERE, "_ref")
LIB = os.path.join(OUT_DIR, "libmoeinf_ref.so")
SRCS = [os.path.join(REF, "core", "parallel", "expert_module.cpp"), os.path.join(REF, "core", "aio", "archer_tensor_index.cpp"),
        os.path.join(HERE, "ref_build", "ref_driver.cpp")]


def available():
    return os.path.exists(LIB)


def needs_build():
    if not os.path.isdir(REF):
        return False  # GPU box: use what travelled
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(s) > t for s in SRCS + [os.path.join(HERE, "ref_build", "cuda_names", "cuda_runtime_api.h")])


def build(force=False, verbose=False):
    if not force and not needs_build():
        return LIB if os.path.exists(LIB) else None
    import pybind11
    import torch

    tdir = os.path.dirname(torch.__file__)
    ti = os.path.join(tdir, "include")
    os.makedirs(OUT_DIR, exist_ok=True)
    cmd = ["g++", "-std=c++17", "-O2", "-fPIC", "-shared", "-w", "-D__HIP_PLATFORM_AMD__", "-DUSE_ROCM",
           f"-D_GLIBCXX_USE_CXX11_ABI={int(torch.compiled_with_cxx11_abi())}",
           "-I" + os.path.join(HERE, "ref_build", "cuda_names"), "-I" + os.path.join(REF, "core"), "-I" + ti,
           "-I" + os.path.join(ti, "torch", "csrc", "api", "include"), "-I" + sysconfig.get_paths()["include"],
           "-I" + pybind11.get_include(), "-I/opt/rocm/include", "-o", LIB] + SRCS + [
           "-L" + os.path.join(tdir, "lib"), "-Wl,-rpath," + os.path.join(tdir, "lib"), "-ltorch", "-ltorch_cpu", "-lc10"]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.run(cmd, check=True)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
