#!/usr/bin/env python3
"""Golden vectors for moe-infinity_amd/priority_score.py, produced by the REFERENCE's own
moe_infinity/memory/expert_priority_score.py (imported from /root/reference, never copied).
TEST INFRASTRUCTURE ONLY.  Writes tests/golden/priority_score.npz.  Re-run: python oracle/gen_golden_priority.py"""
import importlib
import os
import sys
import types

sys.dont_write_bytecode = True
import numpy as np

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "priority_score.npz")


def import_reference():
    for name, path in (("moe_infinity", f"{REF}/moe_infinity"), ("moe_infinity.memory", f"{REF}/moe_infinity/memory")):
        if name not in sys.modules:
            m = types.ModuleType(name)
            m.__path__ = [path]
            m.__package__ = name
            sys.modules[name] = m
    return (importlib.import_module("moe_infinity.memory.expert_priority_score"),
            importlib.import_module("moe_infinity.memory.expert_entry"))


def cases(rng):
    """(L, E, current_layer, expert_freq, decoder matrix, cache entries) — incl. the degenerate branches: no counts in one
    half, an all-zero decoder matrix, zero rows, current layer in either half"""
    out = []
    for L, E in ((12, 8), (8, 16), (24, 4)):
        for cur in (0, L // 2 - 1, L // 2, L - 1):
            for variant in ("dense", "no_decoder_counts", "no_counts", "zero_eam", "zero_rows"):
                freq = {}
                if variant != "no_counts":
                    for l in range(L):
                        if variant == "no_decoder_counts" and l >= L // 2:
                            continue
                        for e in range(E):
                            if rng.random() < 0.6:
                                freq[(e, l)] = int(rng.integers(1, 50))
                eam = rng.integers(0, 6, size=(L, E)).astype(np.float64)
                if variant == "zero_eam":
                    eam[:] = 0
                if variant == "zero_rows":
                    eam[rng.integers(0, L, size=3)] = 0
                cache = [(int(rng.integers(0, E)), int(rng.integers(0, L)), int(rng.integers(0, 1000))) for _ in range(10)]
                out.append((L, E, cur, freq, eam, cache))
    return out


def main():
    ps, ee = import_reference()
    rng = np.random.default_rng(20260926)
    blob = {}
    cs = cases(rng)
    for i, (L, E, cur, freq, eam, cache) in enumerate(cs):
        mat = lambda lst: _matrix(lst, L, E)  # noqa: E731
        blob[f"c{i}_meta"] = np.array([L, E, cur], np.int64)
        blob[f"c{i}_freq"] = np.array([[e, l, v] for (e, l), v in freq.items()], np.int64).reshape(-1, 3)
        blob[f"c{i}_eam"] = eam.copy()
        blob[f"c{i}_cache"] = np.array(cache, np.int64)
        dec = ee.ExpertTraceEntry("s", eam.copy(), 1, 1)
        blob[f"c{i}_priority"] = mat(ps.priority_score(freq, set(), set(), dec, cur, L))
        blob[f"c{i}_eam_after"] = dec.matrix.copy()  # the reference row-normalises the entry's matrix in place
        blob[f"c{i}_oracle"] = mat(ps.oracle_score(freq, ee.ExpertTraceEntry("s", eam.copy(), 1, 1)))
        lf = ps.lfu_score(freq)
        blob[f"c{i}_lfu"] = np.array([[c.expert_idx, c.layer_idx, c.r] for c in lf], np.float64).reshape(-1, 3)
        centries = [ee.ExpertCacheEntry(e, l, 0.0, 0, t) for e, l, t in cache]
        blob[f"c{i}_lru"] = np.array([[c.expert_idx, c.layer_idx, c.r] for c in ps.lru_score(centries)], np.float64)
        blob[f"c{i}_lru_layers"] = np.array([[c.expert_idx, c.layer_idx, c.r] for c in ps.lru_score_with_layers(centries, cur)], np.float64)
    blob["n_cases"] = np.array([len(cs)], np.int64)
    np.savez_compressed(OUT, **blob)
    print("wrote", OUT, len(cs), "cases")


def _matrix(lst, L, E):
    m = np.zeros((L, E), np.float64)
    for c in lst:
        m[c.layer_idx, c.expert_idx] = c.r
    return m


if __name__ == "__main__":
    main()
