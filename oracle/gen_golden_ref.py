"""Golden vectors produced by the REFERENCE'S OWN C++ (oracle/_ref, built by oracle/build_ref.py in a container that
has /root/reference) so that they travel to machines that do not:

  tests/golden/ffn_ref_<family>_<dtype>.npz   x, seeds and y = <reference expert module>.forward(x)
                                              (core/parallel/expert_module.cpp) for a few ragged token counts
  tests/golden/archer_index_ref.bin (+ .json)  an archer_index written by ArcherTensorIndex::Serialize
                                              (core/aio/archer_tensor_index.cpp:101-112) and the entries it holds

TEST INFRASTRUCTURE ONLY.  Re-run:  python oracle/gen_golden_ref.py"""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle import build_ref, ref_lib  # noqa: E402
from oracle.synth import acts, checksum, make_weights  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")
ET = {"mixtral": 4, "deepseek": 5, "nllb": 2, "switch": 0, "fsgpt": 3, "switchgated": 1}  # fsgpt (round 5): NLLB-shaped tensors, its own module (expert_module.cpp:113-129)


def to_np(t):
    return t.float().numpy() if t.dtype in (torch.bfloat16, torch.float16) else t.numpy()


def main():
    build_ref.build()
    for fam in ("mixtral", "deepseek", "nllb", "switch", "fsgpt", "switchgated"):  # switchgated (round 5): DeepSeek-shaped tensors (wi_0, wi_1, wo), gelu gate
        for dt, tag in ((torch.bfloat16, "bf16"), (torch.float32, "f32"), (torch.float16, "f16")):  # f16: round 4 (dtype id 2)
            h, f, e, seed = 256, 352, 3, 4100 + ET[fam]
            kw = {"gate_std": 0.5} if fam in ("nllb", "switch", "fsgpt") else {}
            gate, experts, _ = make_weights({"fsgpt": "nllb", "switchgated": "deepseek"}.get(fam, fam), h, f, e, seed, dt, **kw)
            out = {"meta": np.array([h, f, e, seed]), "wsum": checksum(gate, experts)}
            for i, t in enumerate((1, 5, 37)):
                x = acts(t, h, dt, seed + 10 + i)
                y = ref_lib.expert_ffn(x, experts[i], ET[fam])
                out[f"x{i}"] = to_np(x)
                out[f"y{i}"] = to_np(y)
            np.savez_compressed(os.path.join(GOLD, f"ffn_ref_{fam}_{tag}.npz"), **out)
    entries = {7: dict(file_id=0, offset=0, size=2 * 14336 * 4096, shape=[14336, 4096], dtype=15),
               8: dict(file_id=0, offset=117440512, size=2 * 4096 * 14336, shape=[4096, 14336], dtype=15),
               9: dict(file_id=1, offset=4096, size=4 * 8192, shape=[8192], dtype=6),
               1234567: dict(file_id=3, offset=1 << 33, size=2 * 3, shape=[1, 3, 1], dtype=5),
               10: dict(file_id=0, offset=8192, size=1, shape=[], dtype=11)}
    path = os.path.join(GOLD, "archer_index_ref.bin")
    ref_lib.index_write(path, entries)
    assert ref_lib.index_read(path) == entries
    json.dump({str(k): v for k, v in entries.items()}, open(os.path.join(GOLD, "archer_index_ref.json"), "w"), indent=1)
    print("wrote reference-produced goldens under", GOLD)


if __name__ == "__main__":
    main()
