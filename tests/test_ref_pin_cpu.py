"""Pins the oracle against the REFERENCE'S OWN code (CPU):
  * tests/golden/ffn_ref_*.npz were produced by core/parallel/expert_module.cpp's modules (oracle/_ref, built from
    /root/reference by oracle/build_ref.py; generator oracle/gen_golden_ref.py): the oracle's restated expert FFN must
    reproduce them bit for bit;
  * tests/golden/archer_index_ref.bin was written by ArcherTensorIndex::Serialize (core/aio/archer_tensor_index.cpp):
    the engine's disk tier (csrc/offload_store.h) and the struct-level restatement must read exactly those entries;
  * when oracle/_ref is present (the build container; it also travels to the GPU box) the same is checked live, both
    directions, on fresh random cases."""
import json
import os
import shutil

import numpy as np
import pytest
import torch

from helpers import GOLDEN, R, acts, checksum, load_golden, make_weights, tt
from oracle import offload_format_ref as F
from oracle import ref_lib

ET = {"mixtral": R.MIXTRAL_DENSE_ACT_DENSE, "deepseek": R.DEEPSEEK_DENSE_ACT_DENSE, "nllb": R.NLLB_DENSE_ACT_DENSE,
      "switch": R.SWITCH_DENSE_ACT_DENSE, "fsgpt": R.FSGPT_DENSE_ACT_DENSE,
      "switchgated": R.SWITCH_DENSE_GATED_ACT_DENSE}  # fsgpt: NLLB-shaped tensors; switchgated: (wi_0, wi_1, wo) = DeepSeek-shaped; each its own reference module
CASES = [(fam, dt, tag) for fam in ET for dt, tag in ((torch.bfloat16, "bf16"), (torch.float32, "f32"))]


@pytest.mark.parametrize("fam,dt,tag", CASES)
def test_oracle_ffn_reproduces_the_reference_modules_output(fam, dt, tag):
    z = load_golden(f"ffn_ref_{fam}_{tag}.npz")
    h, f, e, seed = [int(v) for v in z["meta"]]
    gate, experts, _ = make_weights({"fsgpt": "nllb", "switchgated": "deepseek"}.get(fam, fam), h, f, e, seed, dt, **({"gate_std": 0.5} if fam in ("nllb", "switch", "fsgpt") else {}))
    np.testing.assert_array_equal(checksum(gate, experts), z["wsum"])
    for i in range(3):
        x = tt(z[f"x{i}"], dt)
        got = R.expert_ffn(x, experts[i], ET[fam])
        assert torch.equal(got.float(), tt(z[f"y{i}"], torch.float32)), f"{fam} {tag} case {i}: oracle FFN != reference module output"


def test_engine_disk_tier_reads_an_index_written_by_the_reference(tmp_path):
    from moe_infinity_amd.offload_store import OffloadStore

    want = {int(k): v for k, v in json.load(open(os.path.join(GOLDEN, "archer_index_ref.json"))).items()}
    shutil.copy(os.path.join(GOLDEN, "archer_index_ref.bin"), tmp_path / "archer_index")
    got = F.read_index(str(tmp_path / "archer_index"))  # struct-level restatement
    assert {k: (v["file_id"], v["offset"], v["size"], v["shape"], v["dtype"]) for k, v in got.items()} == \
           {k: (v["file_id"], v["offset"], v["size"], v["shape"], v["dtype"]) for k, v in want.items()}
    for v in got.values():  # what offload() of a CPU tensor stores: unpinned, no grad, CPU (-1), strided
        assert (v["pinned"], v["requires_grad"], v["device_index"], v["device_type"], v["layout"]) == (False, False, -1, 0, 0)
    st = OffloadStore(str(tmp_path))  # the engine's reader
    assert sorted(st.ids()) == sorted(want)
    for k, v in want.items():
        m = st.meta(k)
        assert (m["nbytes"], m["offset"], list(m["shape"]), m["scalar_type"]) == (v["size"], v["offset"], v["shape"], v["dtype"])
    st.close()


@pytest.mark.skipif(not ref_lib.available(), reason="oracle/_ref not built (needs /root/reference)")
def test_live_reference_ffn_equals_oracle_on_fresh_cases():
    g = torch.Generator().manual_seed(99)
    for fam, dt, _ in CASES:
        for t in (1, 3, 64):
            h, f = 128 * int(torch.randint(1, 4, (1,), generator=g)), 32 * int(torch.randint(1, 9, (1,), generator=g))
            _, experts, _ = make_weights({"fsgpt": "nllb", "switchgated": "deepseek"}.get(fam, fam), h, f, 1, 500 + t, dt)
            x = acts(t, h, dt, 600 + t)
            assert torch.equal(ref_lib.expert_ffn(x, experts[0], ET[fam]), R.expert_ffn(x, experts[0], ET[fam])), (fam, dt, t, h, f)


@pytest.mark.skipif(not ref_lib.available(), reason="oracle/_ref not built (needs /root/reference)")
def test_live_reference_deserializer_reads_what_the_engine_writes(tmp_path):
    """the other direction: prefetch_handle.offload() semantics through the engine's disk tier, then the REFERENCE'S
    ArcherTensorIndex::Deserialize on the directory's archer_index"""
    from moe_infinity_amd.offload_store import SCALAR_TYPE, OffloadStore

    st = OffloadStore(str(tmp_path))
    tensors = {3: torch.randn(40, 24).to(torch.bfloat16), 4: torch.randn(17), 900: torch.randint(0, 9, (2, 3, 5), dtype=torch.int64)}
    for k, t in tensors.items():
        st.offload(t, k)
    st.close()
    got = ref_lib.index_read(str(tmp_path / "archer_index"))
    assert sorted(got) == sorted(tensors)
    for k, t in tensors.items():
        assert got[k]["shape"] == list(t.shape) and got[k]["size"] == t.numel() * t.element_size() and got[k]["dtype"] == SCALAR_TYPE[t.dtype]
        assert got[k]["offset"] % 4096 == 0
    # and the reference's writer -> the engine's reader, live
    ref_lib.index_write(str(tmp_path / "archer_index"), got)
    st = OffloadStore(str(tmp_path))
    for k, t in tensors.items():
        assert torch.equal(st.load(k), t)
    st.close()
