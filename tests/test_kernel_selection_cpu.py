"""Which form of the register-ring GEMM (csrc/ffn_ring2_kernel.h: ffn_gemm_ring2) an FFN stage takes — the table of DESIGN.md section
4.3, pinned through the introspection export moeinf_ffn_ring2_form (include/moeinf.h; pure host logic, the function the
launchers themselves call: csrc/kernels.h ring2_form).  Why this exists: in round 4 the launcher asked for max_rows <= 192 where
the engine's sync-free path passes 1.5 x the mean + 1 = 193 for a 512-token Mixtral prefill; the new kernel silently never ran
and six GPU experiments measured its predecessor.  No GPU needed."""
import ctypes as C

import pytest

from moe_infinity_amd import load_library

BF16, F32, F16 = 0, 1, 2


def form(dtype, nmat, K, R, active, max_rows, K_sh=0, cus=256):
    lib = load_library()
    out = (C.c_int32 * 5)()
    assert lib.moeinf_ffn_ring2_form(dtype, nmat, K, K_sh, R, active, max_rows, cus, out) == 0
    return tuple(out)  # (token groups per pass, split tail, row blocks per expert, first split unit, workgroups)


def engine_row_estimate(T, K, E):
    """what moeinf_moe_forward passes on the sync-free path (csrc/engine.cpp rows_estimate: min(T, 1.5 T K / E + 1)) — the
    engine's own function, through the C ABI"""
    est = load_library().moeinf_rows_estimate(T, K, E)
    assert est == min(T, (T * K * 3) // (2 * max(1, E)) + 1)
    return est


def test_the_benchmarked_prefill_takes_the_192_token_form_with_a_split_tail():
    est = engine_row_estimate(512, 2, 8)
    assert est == 193
    # Mixtral-8x7B gate-up: 14336 rows = 112 blocks of 128, 8 experts = 896 units = 3.5 rounds of 256 CUs -> the last 128 split
    assert form(BF16, 2, 4096, 14336, 8, est) == (12, 1, 112, 768, 1024)
    # the down projection: 32 blocks x 8 experts = 256 units, one round: nothing to split
    assert form(BF16, 1, 14336, 4096, 8, est) == (12, 0, 32, 0, 256)
    # fp16 experts take the same forms
    assert form(F16, 2, 4096, 14336, 8, est)[:2] == (12, 1)
    assert form(F16, 1, 14336, 4096, 8, est)[:2] == (12, 0)


@pytest.mark.parametrize("tokens,gated,plain", [
    (48, 0, 8),      # 19 rows: the hybrid kernel keeps the gated stage up to 128 rows (<= 16 active experts), ring2 the plain one from 17
    (336, 0, 8),     # 127 rows
    (352, 12, 12),   # 133 rows
    (512, 12, 12),
    (552, 12, 12),   # 208 rows: the last estimate that still takes 192 tokens per pass
    (556, 16, 16),   # 209 rows
    (896, 16, 16),   # 337 rows
    (904, 16, 16),   # 340 rows
    (908, 0, 0),     # 341 rows: the big-tile kernel
    (4096, 0, 0),
])
def test_mixtral_prefill_lengths(tokens, gated, plain):
    est = engine_row_estimate(tokens, 2, 8)
    assert form(BF16, 2, 4096, 14336, 8, est)[0] == gated, est
    assert form(BF16, 1, 14336, 4096, 8, est)[0] == plain, est


def test_row_thresholds():
    # more than 16 active experts: the hybrid kernel stops at 64 rows
    assert form(BF16, 2, 4096, 4096, 32, 64)[0] == 0
    assert form(BF16, 2, 4096, 4096, 32, 65)[0] == 8
    assert form(BF16, 2, 4096, 4096, 32, 128)[0] == 8
    assert form(BF16, 2, 4096, 4096, 32, 129)[0] == 12
    # at most 16: 128 rows
    assert form(BF16, 2, 4096, 14336, 8, 128)[0] == 0
    assert form(BF16, 2, 4096, 14336, 8, 129)[0] == 12
    # fp16 has no hybrid kernel: the gated stage starts at 65 rows whatever the expert count
    assert form(F16, 2, 4096, 14336, 8, 64)[0] == 0
    assert form(F16, 2, 4096, 14336, 8, 65)[0] == 8
    # plain stage: from 17 rows (the decode kernel runs up to 16)
    assert form(BF16, 1, 14336, 4096, 8, 16)[0] == 0
    assert form(BF16, 1, 14336, 4096, 8, 17)[0] == 8
    assert form(BF16, 1, 8192, 2048, 128, 49)[0] == 8  # NLLB-MoE-54B's second stage at a 2048-token prefill (bias epilogue)


def test_shapes_the_ring_does_not_take():
    assert form(BF16, 2, 2048, 1408, 64, 100)[0] == 0               # DeepSeek-V2-Lite: short reduction
    assert form(BF16, 2, 4096 + 32, 14336, 8, 193)[0] == 0          # K not a multiple of 64
    assert form(BF16, 2, 4096, 14336, 8, 193, K_sh=2816)[0] == 0    # a shared expert with a short reduction rides in the same launch
    assert form(BF16, 2, 4096, 14336, 8, 193, K_sh=8192)[0] == 12
    assert form(F32, 2, 4096, 14336, 8, 193)[0] == 0                # fp32 models (Switch): the LDS / hybrid kernels
    assert form(BF16, 1, 2048, 8192, 128, 49)[0] == 0               # NLLB's first stage (K = 2048)


def test_split_tail_rule():
    # only the gated stage, only when the last round fills at most half of the CUs, and only when there is more than one round
    assert form(BF16, 2, 4096, 14336, 8, 193, cus=256)[1:] == (1, 112, 768, 1024)
    assert form(BF16, 2, 4096, 14336, 8, 193, cus=304)[1] == 0      # 896 % 304 = 288 > 152
    assert form(BF16, 2, 4096, 14336, 7, 193, cus=256)[1:] == (1, 112, 768, 800)   # 784 units: 16 left over -> 32 half workgroups
    assert form(BF16, 2, 4096, 4096, 32, 100, cus=256)[1] == 0      # 1024 units: four full rounds
    assert form(BF16, 2, 4096, 2048, 8, 193, cus=256)[1] == 0       # 128 units: less than one round
    assert form(BF16, 1, 14336, 4096 + 128 * 16, 8, 193, cus=256)[1] == 0  # the plain stage is never split
    # every unit is covered exactly once: units below `split` by one workgroup, the others by two
    ntb, tail, nblk, split, blocks = form(BF16, 2, 4096, 14336, 8, 250)
    units = nblk * 8
    assert (ntb, tail) == (16, 1) and blocks == split + 2 * (units - split) and split % 256 == 0


def test_environment_knobs_are_honoured(monkeypatch):
    monkeypatch.setenv("MOEINF_GEMM_RING2", "2")   # plain stage only
    assert form(BF16, 2, 4096, 14336, 8, 193)[0] == 0
    assert form(BF16, 1, 14336, 4096, 8, 193)[0] == 12
    monkeypatch.setenv("MOEINF_GEMM_RING2", "3")
    monkeypatch.setenv("MOEINF_RING2_TAIL", "0")
    assert form(BF16, 2, 4096, 14336, 8, 193)[:2] == (12, 0)
    monkeypatch.setenv("MOEINF_RING2_MAX_ROWS", "256")
    assert form(BF16, 2, 4096, 14336, 8, 257)[0] == 0


def test_the_fence_ring_returns_the_oldest_recorded_fence_that_covers_a_forward():
    """Round 6: sync-free forwards record a fence only every MOEINF_FENCE_EVERY-th time, the decision path after every forward, a
    copy on demand — so the ring holds fences at IRREGULAR forward numbers, and a slot-recycling copy must wait for the oldest
    entry that covers the slot's last reader (csrc/engine_internal.h fence_cover_pos; a younger one is correct but makes the copy
    wait for more compute, an older one is a write under a running kernel).  Against a brute-force model, across wrap-around."""
    import random

    lib = load_library()
    ring = lib.moeinf_fence_ring()
    assert ring >= 16
    rng = random.Random(7)
    seqs = (C.c_uint64 * ring)()
    history = []  # (record index, forward) of every fence ever recorded
    forward = 0
    assert lib.moeinf_fence_cover_pos(seqs, 0, 1) == -1  # nothing recorded yet
    for rec in range(5 * ring):
        forward += rng.choice([1, 1, 1, 2, 16, 16, 32])  # decision-path forwards, on-demand fences, the sync-free cadence
        seqs[rec % ring] = forward
        history.append((rec, forward))
        live = history[-ring:]
        for s in [1, live[0][1] - 1, live[0][1], live[len(live) // 2][1], live[len(live) // 2][1] + 1, forward - 1, forward, forward + 1]:
            if s < 1:
                continue
            want = next((r % ring for r, f in live if f >= s), -1)
            assert lib.moeinf_fence_cover_pos(seqs, rec + 1, s) == want, (rec, s, forward)
