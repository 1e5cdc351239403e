"""Pin the oracle (oracle/moe_ref.py) against golden vectors produced by the reference's
own Python blocks (oracle/gen_golden.py).  CPU only."""
import os

import numpy as np
import pytest
import torch

from oracle import moe_ref as R
from oracle.synth import checksum, make_weights

G = os.path.join(os.path.dirname(__file__), "golden")


def load(name):
    return np.load(os.path.join(G, name), allow_pickle=False)


def t(a, dtype):
    return torch.from_numpy(np.asarray(a)).to(dtype)


def assert_close_model_dtype(got, ref, dtype, what):
    """bf16 tensors may differ by rounding flips where the reference's ATen accumulation
    order differs from ours: allow <= 1 bf16 ulp on any element, and require the bulk
    (mean abs error) to be far below 1e-3 of the output scale."""
    got, ref = got.float(), ref.float()
    err = (got - ref).abs()
    scale = ref.abs().max().item() + 1e-30
    if dtype == torch.float32:
        assert err.max().item() <= 2e-5 * scale, (what, err.max().item(), scale)
    else:
        # 1 bf16 ulp of the element, or of the tensor's typical magnitude where terms cancel
        ulp = torch.maximum(torch.maximum(ref.abs(), got.abs()), ref.abs().mean()) * 2.0 ** -7 + 1e-30
        assert bool((err <= ulp).all()), (what, (err / ulp).max().item())
        assert err.mean().item() <= 1e-3 * ref.abs().mean().item(), (what, err.mean().item())


@pytest.mark.parametrize("name", ["mixtral_decode_b1.npz", "mixtral_decode_b4.npz", "mixtral_prefill_t48.npz"])
def test_mixtral_block_matches_reference(name):
    z = load(name)
    b, s, h, f, e, k, seed = [int(v) for v in z["meta"]]
    gate, experts, _ = make_weights("mixtral", h, f, e, seed, torch.bfloat16)
    np.testing.assert_allclose(checksum(gate, experts), z["wsum"], rtol=0, atol=0)
    x = t(z["x"], torch.bfloat16)
    r = R.block_mixtral(x, gate, experts, top_k=k)
    # routing indices: bit-exact vs the reference (torch.topk); golden seeds are tie-free
    assert torch.equal(r.topk_idx, t(z["topk_idx"], torch.int64)), name
    assert torch.equal(r.logits.float(), t(z["logits"], torch.float32)), "gate logits (bf16) must be bit-equal"
    assert_close_model_dtype(r.topk_w, t(z["topk_w"], torch.float32), torch.bfloat16, "topk_w")
    assert_close_model_dtype(r.out, t(z["out"], torch.float32), torch.bfloat16, "out")


@pytest.mark.parametrize("name", ["grok_decode_b1.npz", "grok_prefill_t40.npz"])
def test_grok_block_matches_reference(name):
    """SyncGrokMoeBlock (moe_infinity/models/grok.py:34-95): softmax -> top-2 WITHOUT renormalisation; experts as the reference's
    core runs them for this architecture (type 4 over the blob in named_parameters order; oracle/gen_golden.py gen_grok)."""
    z = load(name)
    b, s, h, f, e, k, seed = [int(v) for v in z["meta"]]
    gate, experts, _ = make_weights("mixtral", h, f, e, seed, torch.bfloat16)
    np.testing.assert_allclose(checksum(gate, experts), z["wsum"], rtol=0, atol=0)
    x = t(z["x"], torch.bfloat16)
    r = R.block_grok(x, gate, experts, top_k=k)
    assert torch.equal(r.topk_idx, t(z["topk_idx"], torch.int64)), name
    assert torch.equal(r.logits.float(), t(z["logits"], torch.float32)), "gate logits (bf16) must be bit-equal"
    assert torch.equal(r.topk_w.float(), t(z["topk_w"], torch.float32)), "un-renormalised weights are a cast of the softmax: bit-equal"
    assert float(r.topk_w.float().sum(-1).max()) < 0.999, "the weights of a token do not sum to one (no renormalisation)"
    assert_close_model_dtype(r.out, t(z["out"], torch.float32), torch.bfloat16, "out")


@pytest.mark.parametrize("name", ["deepseek_decode_b1.npz", "deepseek_prefill_t40.npz", "deepseek_group_t16.npz"])
def test_deepseek_block_matches_reference(name):
    z = load(name)
    b, s, h, f, e, k, n_shared, seed = [int(v) for v in z["meta"]]
    method, n_group, topk_group, norm, scaling = [str(v) for v in z["cfg"]]
    gate, experts, shared = make_weights("deepseek", h, f, e, seed, torch.bfloat16, n_shared=n_shared)
    np.testing.assert_allclose(checksum(gate, experts, shared), z["wsum"], rtol=0, atol=0)
    x = t(z["x"], torch.bfloat16)
    kw = dict(topk_method=method, norm_topk_prob=bool(int(norm)), routed_scaling_factor=float(scaling))
    if method != "greedy":
        kw.update(n_group=int(n_group), topk_group=int(topk_group))
    r = R.block_deepseek(x, gate, experts, k, shared=shared, **kw)
    # sorted=False in the reference: compare per-token sets, weights keyed by expert id
    ref_idx, ref_w = t(z["topk_idx"], torch.int64), t(z["topk_w"], torch.float32)
    for tok in range(ref_idx.shape[0]):
        ref_map = {int(i): float(w) for i, w in zip(ref_idx[tok], ref_w[tok])}
        got_map = {int(i): float(w) for i, w in zip(r.topk_idx[tok], r.topk_w[tok])}
        assert set(ref_map) == set(got_map), (name, tok)
        for i in ref_map:
            assert abs(ref_map[i] - got_map[i]) <= 2e-6 * max(1.0, abs(ref_map[i])), (name, tok, i)
    assert_close_model_dtype(r.out, t(z["out"], torch.float32), torch.bfloat16, "out")


@pytest.mark.parametrize("name", ["deepseekv3_decode_b1.npz", "deepseekv3_prefill_t40.npz", "deepseekv3_e256_t24.npz"])
def test_deepseek_v3_gate_block_matches_reference(name):
    """DeepseekMoEBlock with the V3 gate (modeling_deepseek_v3 MoEGate: sigmoid scores, e_score_correction_bias, top-2-sum group
    selection, weights normalised then scaled): the reference block's own output and routing (oracle/gen_golden.py gen_deepseek_v3)."""
    z = load(name)
    b, s, h, f, e, k, n_shared, seed = [int(v) for v in z["meta"]]
    method, n_group, topk_group, norm, scaling = [str(v) for v in z["cfg"]]
    assert method == "noaux_tc"
    gate, experts, shared = make_weights("deepseek", h, f, e, seed, torch.bfloat16, n_shared=n_shared)
    np.testing.assert_allclose(checksum(gate, experts, shared), z["wsum"], rtol=0, atol=0)
    x = t(z["x"], torch.bfloat16)
    r = R.block_deepseek(x, gate, experts, k, shared=shared, e_bias=t(z["e_bias"], torch.float32), n_group=int(n_group), topk_group=int(topk_group),
                         norm_topk_prob=bool(int(norm)), routed_scaling_factor=float(scaling))
    ref_idx = t(z["topk_idx"], torch.int64)
    assert torch.equal(r.topk_idx.sort(-1).values, ref_idx.sort(-1).values), "routing sets must be bit-exact"
    # weights: compare as idx -> weight maps (sorted=False in the reference)
    got = torch.zeros(ref_idx.shape[0], e).scatter_(1, r.topk_idx, r.topk_w.float())
    want = torch.zeros(ref_idx.shape[0], e).scatter_(1, ref_idx, t(z["topk_w"], torch.float32))
    assert torch.allclose(got, want, rtol=1e-6, atol=1e-7)
    assert_close_model_dtype(r.out, t(z["out"], torch.float32), torch.bfloat16, "out")


@pytest.mark.parametrize("name", ["switch_decode_b1.npz", "switch_prefill_cap.npz"])
def test_switch_block_matches_reference(name):
    z = load(name)
    b, s, h, f, e, cap, seed = [int(v) for v in z["meta"]]
    gate, experts, _ = make_weights("switch", h, f, e, seed, torch.float32, gate_std=0.5)
    np.testing.assert_allclose(checksum(gate, experts), z["wsum"], rtol=0, atol=0)
    x = t(z["x"], torch.float32)
    r = R.block_switch(x, gate, experts, expert_capacity=cap)
    assert torch.equal(r.router_mask.reshape(b, s, e).to(torch.uint8), t(z["router_mask"], torch.uint8))
    np.testing.assert_allclose(r.extra["router_probs"].numpy(), z["router_probs"], rtol=2e-5, atol=1e-7)  # fp32 gate accumulate order differs
    assert_close_model_dtype(r.out, t(z["out"], torch.float32), torch.float32, "out")


@pytest.mark.parametrize("name,dtype", [("nllb_decode_b8.npz", torch.bfloat16), ("nllb_prefill_f32.npz", torch.float32)])
def test_nllb_block_matches_reference(name, dtype):
    z = load(name)
    b, s, h, f, e, seed, norm_before = [int(v) for v in z["meta"]]
    gate, experts, _ = make_weights("nllb", h, f, e, seed, dtype, gate_std=0.5)
    np.testing.assert_allclose(checksum(gate, experts), z["wsum"], rtol=0, atol=0)
    x = t(z["x"], dtype)
    r = R.block_nllb(x, gate, experts, normalize_router_prob_before_dropping=bool(norm_before))
    ref_probs = t(z["router_probs"], torch.float32)
    assert torch.equal(r.weights_mask.bool(), ref_probs.bool()), "routing sets must be bit-exact"
    assert torch.equal(torch.argmax(r.extra["top_1_mask"], dim=-1), t(z["top1"], torch.int64))
    assert_close_model_dtype(r.weights_mask, ref_probs, dtype, "combining weights")
    assert_close_model_dtype(r.out, t(z["out"], torch.float32), dtype, "out")


def test_topk_rule_equals_torch_topk_on_tie_free_rows():
    g = torch.Generator().manual_seed(0)
    p = torch.rand(512, 64, generator=g)
    v, i = R.topk_lowest_index(p, 6)
    tv, ti = torch.topk(p, 6, dim=-1)
    assert torch.equal(i, ti) and torch.equal(v, tv)
    # ties: lowest index first
    q = torch.tensor([[0.5, 0.5, 0.1, 0.5]])
    assert R.topk_lowest_index(q, 2)[1].tolist() == [[0, 1]]


def test_dispatch_index_order():
    mask = torch.tensor([[1, 0, 1], [0, 0, 1], [1, 0, 0], [1, 0, 1]], dtype=torch.bool)
    counts, offsets, slot_token, experts = R.dispatch_index(mask)
    assert counts.tolist() == [3, 0, 3] and offsets.tolist() == [0, 3, 3, 6]
    assert slot_token.tolist() == [0, 2, 3, 0, 1, 3] and experts == [0, 2]
