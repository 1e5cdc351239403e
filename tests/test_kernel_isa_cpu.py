"""Properties of the generated gfx950 ISA that performance depends on and that a source refactoring can silently lose (hipcc
cross-compiles here without a GPU).  Round 6: the tier mover's pull kernel dropped from 56 to 42 GB/s when a register zeroed in
front of a predicated load made the compiler put ``s_waitcnt vmcnt(0)`` in front of every load — one request per lane in flight
instead of four — and nothing but a GPU A/B noticed."""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = "/opt/rocm/bin/hipcc"


@pytest.fixture(scope="module")
def kernels_asm(tmp_path_factory):
    if not os.path.exists(HIPCC):
        pytest.skip("no hipcc")
    out = tmp_path_factory.mktemp("isa") / "kernels.s"
    subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only", "-o", str(out),
                    os.path.join(ROOT, "moe-infinity_amd", "csrc", "kernels.hip")], check=True, capture_output=True, timeout=600)
    return open(out).read().split("\n")


def _kernel(lines, mangled_prefix):
    start = next(i for i, l in enumerate(lines) if l.startswith(mangled_prefix) and l.rstrip().endswith(":") or l.startswith(mangled_prefix + "E") and ":" in l)
    end = next(i for i in range(start, len(lines)) if "s_endpgm" in lines[i])
    return lines[start:end]


@pytest.mark.parametrize("inst", ["ItLb0EE", "ItLb1EE", "IfLb0EE"], ids=["bf16_or_fp16", "fp8_source", "fp32"])
def test_pull_kernel_keeps_four_loads_per_lane_in_flight(kernels_asm, inst):
    body = _kernel(kernels_asm, "_ZN6moeinf18pull_retile_kernel" + inst)
    ops = [(i, l.strip()) for i, l in enumerate(body) if re.search(r"global_load_dwordx4|s_waitcnt.*vmcnt\(\d+\)", l)]
    loads = [k for k, (_, l) in enumerate(ops) if "global_load_dwordx4" in l]
    assert len(loads) == 8, f"prologue + in-loop: two batches of four host loads expected, found {len(loads)}"
    for batch in (loads[:4], loads[4:]):
        assert batch == list(range(batch[0], batch[0] + 4)), "a wait sits between the four loads of a unit:\n" + "\n".join(l for _, l in ops)
    assert all("nt" in ops[k][1] for k in loads), "the host blob is read once: non-temporal loads"
    # no register spills in the copy loop
    assert not any("scratch_" in l for l in body)
