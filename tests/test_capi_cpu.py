"""CPU-side checks of the C-ABI library: it loads, exports every symbol include/moeinf.h
declares, and its host-only components (cache policy, tracer) agree with the oracle."""
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    import __graft_entry__ as g

    g.build()
    from moe_infinity_amd import load_library

    return load_library()


def test_header_symbols_are_exported(lib):
    hdr = open(os.path.join(ROOT, "include", "moeinf.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(moeinf_[a-z_0-9]+)\s*\(", hdr))
    assert len(declared) >= 30
    from moe_infinity_amd._lib import PROTOTYPES

    assert declared == set(PROTOTYPES), declared ^ set(PROTOTYPES)
    for name in declared:
        assert hasattr(lib, name), f"libmoeinf_hip.so does not export {name}"
    assert lib.moeinf_abi_version() == 4


def test_struct_sizes_match_header():
    import ctypes as C

    from moe_infinity_amd._lib import Config, Stats

    # 17 int32/float fields + alignment + double + 2 int64 + 4 int32 (see include/moeinf.h)
    assert C.sizeof(Config) == 17 * 4 + 4 + 8 + 16 + 16
    assert C.sizeof(Stats) == 23 * 8


def test_no_gpu_fails_loudly():
    import torch

    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from moe_infinity_amd import MoEEngine
    from moe_infinity_amd import config as Cf

    with pytest.raises(RuntimeError, match="no CPU fallback"):
        MoEEngine(Cf.mixtral_8x7b())


def test_cache_policy_matches_oracle(lib):
    from moe_infinity_amd import CacheSim
    from oracle.cache_ref import RefCache

    rng = np.random.default_rng(0)
    for policy in (0, 1):
        for slots in (1, 3, 8):
            sim, ref = CacheSim(slots, policy), RefCache(slots, policy)
            for step in range(2000):
                if step % 500 == 250:
                    sim.clear_counts()
                    ref.clear_counts()
                if step % 300 == 100:
                    prot = rng.choice(24, size=min(slots, 2), replace=False).tolist()
                    sim.protect(prot)
                    ref.protect(prot)
                i = int(rng.zipf(1.3) % 24)
                assert sim.access(i) == ref.access(i), (policy, slots, step)


def test_tracer_matches_reference_golden(lib):
    """moe_infinity/memory/expert_{tracer,predictor,prefetcher}.py run on CPU by oracle/gen_golden.py."""
    from moe_infinity_amd import ExpertTracerNative

    for name in ("tracer_l6_e8_full.npz", "tracer_l6_e8_partial.npz"):
        z = np.load(os.path.join(ROOT, "tests", "golden", name))
        L, E, cap, n_hist, steps, k, seed = [int(v) for v in z["meta"]]
        tr = ExpertTracerNative(L, E, cap)
        tr.load_trace(z["hist"][:n_hist])
        seq = tr.create_entry()
        call = 0
        for st in range(steps):
            for l in range(L):
                m, nearest = tr.predict(seq, z["sel"][call], l)
                assert nearest == int(z["nearest"][call]), (name, call)
                np.testing.assert_array_equal(m, z["pred"][call].astype(np.float32))
                ls, es, sc = tr.prefetch_order(l, m)
                want = z["order"][call]
                want = want[want >= 0]
                assert np.array_equal(ls * E + es, want), (name, call)
                call += 1
        np.testing.assert_array_equal(tr.get_eam(seq), z["eam"])


def test_prefetch_queue_matches_the_restated_task_pool(lib):
    """csrc/prefetch_queue.h (the engine's pending-transfer queue) against oracle/prefetch_queue_ref.py (the reference's
    ArcherTaskPool queue discipline), operation by operation on random traces: identical displaced counts, pop order
    and queue content."""
    import ctypes as C
    import random

    from oracle.prefetch_queue_ref import RefTaskQueue

    rng = random.Random(7)
    for trial in range(20):
        h = C.c_void_p()
        assert lib.moeinf_pq_create(C.byref(h)) == 0
        ref = RefTaskQueue()
        d = C.c_int32()
        for step in range(300):
            op = rng.random()
            node, layer = rng.randrange(24), 0
            layer = node // 4  # 4 experts per layer
            if op < 0.45:
                pr, rl = rng.choice([1, 1, 1, 2, 5, 19]), rng.random() < 0.15
                assert lib.moeinf_pq_enqueue(h, node, layer, pr, int(rl), C.byref(d)) == 0
                assert d.value == ref.enqueue(node, layer, pr, rl)
            elif op < 0.6:
                nd = node if rng.random() < 0.7 else -1
                assert lib.moeinf_pq_on_demand(h, nd, layer, C.byref(d)) == 0
                assert d.value == ref.on_demand(nd, layer)
            elif op < 0.7:
                there = rng.random() < 0.3
                assert lib.moeinf_pq_fetch(h, node, layer, int(there), C.byref(d)) == 0
                assert d.value == ref.fetch(node, layer, there)
            elif op < 0.73:
                assert lib.moeinf_pq_clear_prefetch(h, C.byref(d)) == 0
                assert d.value == ref.clear_prefetch()
            else:
                n, l, p, f = C.c_int64(), C.c_int32(), C.c_int32(), C.c_int32()
                assert lib.moeinf_pq_pop(h, C.byref(n), C.byref(l), C.byref(p), C.byref(f)) == 0
                want = ref.pop()
                assert bool(f.value) == (want is not None)
                if want is not None:
                    assert (n.value, l.value, p.value) == want
            nodes = (C.c_int64 * 512)()
            layers = (C.c_int32 * 512)()
            prios = (C.c_int32 * 512)()
            cnt = C.c_int32()
            assert lib.moeinf_pq_snapshot(h, nodes, layers, prios, 512, C.byref(cnt)) == 0
            assert [(nodes[i], layers[i], prios[i]) for i in range(cnt.value)] == ref.snapshot()
        lib.moeinf_pq_destroy(h)


def test_priority_from_score_orders_levels(lib):
    import ctypes as C

    lv = C.c_int32()
    levels = []
    for s in (1.0, 0.9, 0.5, 0.2, 0.05, 0.0, -1.0):
        assert lib.moeinf_priority_from_score(C.c_float(s), C.byref(lv)) == 0
        levels.append(lv.value)
    assert levels == sorted(levels) and levels[0] == 1 and levels[-1] == 19 and all(1 <= v <= 19 for v in levels)
