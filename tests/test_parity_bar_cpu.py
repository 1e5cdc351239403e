"""The parity bar itself (oracle/parity.py), on CPU: the NLLB `next_states == 0` passthrough
(moe_infinity/models/nllb_moe.py:103) is a discontinuity — a one-ulp flip in an expert output moves the block
output by a full-size |x| where two weighted contributions cancel exactly.  The bar must accept both sides of that
discontinuity there and nothing else."""
import torch

from helpers import P, R, acts, make_weights


def _nllb_block(t=48, h=256, f=128, e=8, seed=5):
    gate, experts, _ = make_weights("nllb", h, f, e, seed, torch.bfloat16, gate_std=0.5)
    x = acts(t, h, torch.bfloat16, seed + 1)[None]
    return x, R.block_nllb(x, gate, experts)


def test_one_ulp_flip_at_an_exact_cancellation_is_a_full_size_difference_and_the_bar_names_it():
    x, ref = _nllb_block()
    pre = ref.extra["pre_passthrough"].float()
    # force an exact cancellation at one element, as bf16 sums of two opposite-sign contributions produce
    # about once in a few hundred elements at full size
    t, c = 3, 17
    ref.extra["pre_passthrough"][0, t, c] = 0.0
    ref.out[0, t, c] = x[0, t, c]  # what the reference returns there: the INPUT element
    got = ref.out.clone()
    # the other side of the discontinuity: the kernel's sum was one flip away from 0 -> a tiny value, no passthrough
    mag = P.block_magnitude(ref)[0, t, c].item()
    got[0, t, c] = mag * 2.0 ** -8
    naive = (got.float() - ref.out.float()).abs()[0, t, c] / (max(abs(float(ref.out[0, t, c])), 1e-9) * 2.0 ** -7)
    assert naive > 50, "without the rule this element is ~128x out of a one-ulp bar"
    rep = P.block_report(got, ref, torch.bfloat16)
    assert rep["ok"] and rep["passthrough_ambiguous"] >= 1, rep
    # and the reverse: oracle sum tiny but non-zero, kernel hit the exact 0 and passed x through
    ref.extra["pre_passthrough"][0, t, c] = mag * 2.0 ** -8
    ref.out[0, t, c] = ref.extra["pre_passthrough"][0, t, c]
    got[0, t, c] = x[0, t, c]
    assert P.block_report(got, ref, torch.bfloat16)["ok"]
    assert float(pre.abs().max()) > 0


def test_the_rule_does_not_excuse_real_errors():
    x, ref = _nllb_block()
    got = ref.out.clone()
    # a wrong value at an element whose sum is nowhere near 0
    pre = ref.extra["pre_passthrough"].float()
    t, c = [int(v) for v in (pre[0].abs() == pre[0].abs().max()).nonzero()[0]]
    got[0, t, c] = got[0, t, c] * 1.5
    rep = P.block_report(got, ref, torch.bfloat16)
    assert not rep["ok"] and rep["n_bad"] == 1
    # a near-zero element that is neither the small sum nor the input element
    ref.extra["pre_passthrough"][0, 1, 1] = 0.0
    ref.out[0, 1, 1] = x[0, 1, 1]
    got = ref.out.clone()
    got[0, 1, 1] = 0.37
    assert not P.block_report(got, ref, torch.bfloat16)["ok"]


def test_rows_bar_is_one_ulp():
    a = torch.randn(64, 64).to(torch.bfloat16)
    b = a.clone()
    b[0, 0] = (b[0, 0].float() * (1 + 2.0 ** -8)).to(torch.bfloat16)  # at most one bf16 step
    assert P.rows_report(b, a, torch.bfloat16)["ok"]
    b[1, 1] = b[1, 1].float() * 1.1 + 0.5
    assert not P.rows_report(b, a, torch.bfloat16)["ok"]


def test_the_fp32_exact_arm_accepts_the_oracle_and_rejects_a_lossy_kernel():
    """oracle/parity.py exact_block / accuracy_report: a result as accurate as the reference's CPU path passes; one that rounds
    the FFN's hidden activations to bf16 a second time with TRUNCATION (a kernel that loses half a bit) stays inside a few ulps
    elementwise yet is measurably further from the fp32 computation — the arm names it."""
    import torch.nn.functional as F

    for family in ("mixtral", "deepseek", "nllb", "switch"):
        dt = torch.float32 if family == "switch" else torch.bfloat16
        e, k, n_shared = (16, 4, 2) if family == "deepseek" else (8, 2, 0)
        gate, experts, shared = make_weights(family, 256, 512, e, 77, dt, n_shared=n_shared, **({"gate_std": 0.5} if family in ("nllb", "switch") else {}))
        x = acts(24, 256, dt, 78)[None]
        if family == "mixtral":
            ref = R.block_mixtral(x, gate, experts, top_k=k)
        elif family == "deepseek":
            ref = R.block_deepseek(x, gate, experts, k, shared=shared)
        elif family == "nllb":
            ref = R.block_nllb(x, gate, experts)
        else:
            ref = R.block_switch(x, gate, experts, expert_capacity=64)
        ex = P.exact_block(family, x, ref, experts, shared=shared)
        rep = P.accuracy_report(ref.out, ref, ex, dt)
        assert rep["ok"] and abs(rep["ratio"] - 1.0) < 1e-9 and rep["oracle_vs_exact"] > 0, (family, rep)
        # rows arm, same identity
        order = sorted(ref.expert_out)
        rows_ref = torch.cat([ref.expert_out[i] for i in order], 0)
        assert P.rows_accuracy_report(rows_ref, rows_ref, torch.cat([ex["rows"][i] for i in order], 0))["ok"]
    # the lossy kernel (Mixtral): hidden activations truncated instead of rounded
    gate, experts, _ = make_weights("mixtral", 256, 512, 8, 79, torch.bfloat16)
    x = acts(64, 256, torch.bfloat16, 80)[None]
    ref = R.block_mixtral(x, gate, experts, top_k=2)
    ex = P.exact_block("mixtral", x, ref, experts)
    trunc = lambda t: (t.float().view(torch.int32) & ~0xFFFF).view(torch.float32).to(torch.bfloat16)  # noqa: E731
    lossy = torch.zeros_like(ref.out[0])
    for i in sorted(ref.expert_out):  # the oracle's op sequence (expert_module.cpp:147-175 in bf16) with truncation at two rounding points
        tok = ref.router_mask[:, i].bool()
        w1, w2, w3 = experts[i]
        xi = x[0][tok]
        hcur = trunc(F.silu(torch.matmul(xi, w1.t())).float() * torch.matmul(xi, w3.t()).float())
        y = trunc(torch.matmul(hcur.float(), w2.float().t()))
        lossy[tok] += y * ref.weights_mask[tok, i][:, None]
    assert P.block_report(lossy[None], ref, torch.bfloat16)["worst"] < 4.0  # a few ulps elementwise: the ulp bars barely notice
    rep = P.accuracy_report(lossy[None], ref, ex, torch.bfloat16)
    assert not rep["ok"] and rep["ratio"] > 1.15, rep
