"""moe-infinity_amd/priority_score.py against golden vectors produced by the reference's own
moe_infinity/memory/expert_priority_score.py (oracle/gen_golden_priority.py).  Host-only."""
import os

import numpy as np

from moe_infinity_amd import priority_score as PS

Z = np.load(os.path.join(os.path.dirname(__file__), "golden", "priority_score.npz"))


def _case(i):
    L, E, cur = [int(v) for v in Z[f"c{i}_meta"]]
    freq = {(int(e), int(l)): int(v) for e, l, v in Z[f"c{i}_freq"]}
    return L, E, cur, freq, Z[f"c{i}_eam"].copy(), [tuple(int(v) for v in row) for row in Z[f"c{i}_cache"]]


def test_priority_score_matches_the_reference_bit_for_bit():
    n = int(Z["n_cases"][0])
    assert n >= 60
    for i in range(n):
        L, E, cur, freq, eam, _ = _case(i)
        dec = PS.ExpertTraceEntry("s", eam, 1, 1)
        got = PS.score_matrix(PS.priority_score(freq, set(), set(), dec, cur, L), L, E)
        assert np.array_equal(got, Z[f"c{i}_priority"]), f"case {i}"
        assert np.array_equal(dec.matrix, Z[f"c{i}_eam_after"]), f"case {i}: in-place row normalisation of the entry"
        keep = eam.copy() if False else Z[f"c{i}_eam"].copy()
        dec2 = PS.ExpertTraceEntry("s", keep, 1, 1)
        got2 = PS.score_matrix(PS.priority_score(freq, set(), set(), dec2, cur, L, inplace=False), L, E)
        assert np.array_equal(got2, Z[f"c{i}_priority"]) and np.array_equal(dec2.matrix, Z[f"c{i}_eam"])


def test_oracle_lfu_lru_scores_match_the_reference():
    for i in range(int(Z["n_cases"][0])):
        L, E, cur, freq, eam, cache = _case(i)
        got = PS.score_matrix(PS.oracle_score(freq, PS.ExpertTraceEntry("s", eam, 1, 1)), L, E)
        assert np.array_equal(got, Z[f"c{i}_oracle"])
        lf = np.array([[c.expert_idx, c.layer_idx, c.r] for c in PS.lfu_score(freq)], np.float64).reshape(-1, 3)
        assert np.array_equal(lf, Z[f"c{i}_lfu"])
        ce = [PS.ExpertCacheEntry(e, l, 0.0, 0, t) for e, l, t in cache]
        assert np.array_equal(np.array([[c.expert_idx, c.layer_idx, c.r] for c in PS.lru_score(ce)], np.float64), Z[f"c{i}_lru"])
        assert np.array_equal(np.array([[c.expert_idx, c.layer_idx, c.r] for c in PS.lru_score_with_layers(ce, cur)], np.float64),
                              Z[f"c{i}_lru_layers"])


def test_levels_follow_the_engines_score_map():
    from moe_infinity_amd import load_library
    import ctypes as C

    lib = load_library()
    scores = np.array([1.0, 0.9, 0.5, 0.26, 0.05, 1e-9, 0.0])
    lv = PS.levels_from_scores(scores)
    assert lv[0] == 1 and lv[-1] == 19 and np.all(np.diff(lv) >= 0)
    for s, want in zip(scores / scores.max(), lv):
        out = C.c_int32(0)
        assert lib.moeinf_priority_from_score(C.c_float(float(s)), C.byref(out)) == 0
        assert out.value == int(want), (s, out.value, want)


def test_prefetcher_orders_requests_by_priority_score():
    from moe_infinity_amd.memory import ExpertPrefetcher

    class FakeTracer:
        _native = None

    class FakeEngine:
        def __init__(self):
            self.protected, self.calls = None, []

        def protect(self, pairs):
            self.protected = list(pairs)

        def prefetch(self, layer, experts, scores=None):
            self.calls.append((layer, list(experts), list(scores)))

    L, E = 8, 4
    pf = ExpertPrefetcher(L, E, FakeTracer())
    eng = FakeEngine()
    pf.set_archer_engine(eng)
    rng = np.random.default_rng(3)
    freq = {(e, l): int(rng.integers(1, 9)) for l in range(L) for e in range(E)}
    eam = rng.integers(0, 5, size=(L, E)).astype(np.float64)
    scored = pf.prefetch_experts_by_priority(2, freq, eam, max_experts=6)
    assert len(scored) == 6 and all(c.layer_idx > 2 for c in scored)
    assert [c.r for c in scored] == sorted((c.r for c in scored), reverse=True)
    assert eng.protected == [(c.layer_idx, c.expert_idx) for c in scored]
    flat = [(l, e, s) for l, es, ss in eng.calls for e, s in zip(es, ss)]
    assert [(l, e) for l, e, _ in flat] == eng.protected and flat[0][2] == 1.0 and all(0 < s <= 1.0 for _, _, s in flat)
    want = PS.score_matrix(PS.priority_score(freq, set(), set(), PS.ExpertTraceEntry("s", eam.copy()), 2, L, inplace=False), L, E)
    assert all(abs(c.r - want[c.layer_idx, c.expert_idx]) == 0 for c in scored)
