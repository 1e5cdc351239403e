"""Model-level (chained) parity through the boundary (protocol of
/root/reference/examples/interface_example.py:120-156: greedy decode, one new token per step).

A small decoder is built around L MoE layers: embedding -> L x [h += Lin_l(rmsnorm(h)); h += MoE_l(rmsnorm(h))] ->
rmsnorm -> lm_head -> argmax.  Everything outside the MoE blocks (norms, the fixed linear "attention" stand-in, the
lm_head) is computed ONCE, on the CPU, by the same code for both arms, so the only difference between the arms is the
path under test: the oracle's block (oracle/moe_ref.py) vs the HIP engine through the C ABI, where layer l's GPU output
feeds layer l+1 and every step's argmax feeds the next step.

  * teacher-forced arm: every GPU layer gets the ORACLE's input of that layer -> routing indices must be bit-exact
    and the block output inside the block bar, for every (step, layer);
  * free-running arm: the GPU arm runs on its own outputs for L layers x 32 steps.  Per step (while the two arms share
    the input token) the mean-relative logit error |gpu - oracle| / mean|oracle| is measured; its median over the steps
    must be <= 1e-3 (north_star's tolerance) and no step may exceed one bf16 ulp (2^-8 = 3.9e-3): the residual stream
    is re-rounded to bf16 after every block, so a last-bit difference in a block output can flip a rounding of h (ulp
    of h ~ 30x the ulp of the block output) — the same happens between any two correct bf16 implementations;
  * accuracy arm: the same chain in fp32 (oracle blocks on fp32 copies of the weights, fp32 residual stream) is the
    "exact" answer; the GPU chain must be as close to it as the reference's CPU path (the bf16 oracle chain) is:
    mean |gpu - exact| <= 1.15 x mean |oracle - exact|.  Token agreement is reported and must be near-total.
"""
import pytest
import torch

from helpers import R, assert_block_close, engine_for, make_weights, register_all

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
L, H, V, STEPS = 4, 512, 640, 32


def _rms(h):
    hf = h.float()
    return (hf / hf.pow(2).mean(-1, keepdim=True).add(1e-6).sqrt()).to(torch.bfloat16)


class _Decoder:
    """the dense parts of the toy decoder (CPU, fp32 math, bf16 activations) — shared by both arms"""

    def __init__(self, seed):
        g = torch.Generator().manual_seed(seed)
        self.emb = torch.randn(V, H, generator=g).to(torch.bfloat16)
        self.pos = (torch.randn(STEPS, H, generator=g) * 0.5).to(torch.bfloat16)  # so a repeated token still gives a new input
        self.lin = [(torch.randn(H, H, generator=g) * (0.5 / H ** 0.5)).to(torch.bfloat16) for _ in range(L)]
        self.lm = (torch.randn(V, H, generator=g) / H ** 0.5).to(torch.bfloat16)

    def pre_moe(self, h, l):
        """h -> (h after the linear 'attention' residual, the MoE block's input)"""
        h = (h.float() + _rms(h).float() @ self.lin[l].float().T).to(torch.bfloat16)
        return h, _rms(h)

    def logits(self, h):
        return _rms(h).float() @ self.lm.float().T


def _family_setup(family):
    if family == "mixtral":
        f, e, k, n_shared = 1024, 8, 2, 0
    else:
        f, e, k, n_shared = 256, 16, 4, 2
    ws = [make_weights(family, H, f, e, 9100 + 13 * l, torch.bfloat16, n_shared=n_shared) for l in range(L)]
    eng = engine_for(family, H, f, e, k, torch.bfloat16, n_shared=n_shared, max_tokens=4, num_layers=L)
    for l in range(L):
        register_all(eng, ws[l][1], ws[l][2], layer=l)
    return ws, eng, e, k


def _oracle_block(family, x, w, k):
    gate, experts, shared = w
    if family == "mixtral":
        return R.block_mixtral(x[None], gate, experts, top_k=k)
    return R.block_deepseek(x[None], gate, experts, k, shared=shared)


@pytest.mark.parametrize("family", ["mixtral", "deepseek"])
def test_chained_greedy_decode_teacher_forced_and_free_running(family):
    ws, eng, e, k = _family_setup(family)
    gates = [w[0].to(DEV) for w in ws]
    dec = _Decoder(77)
    # warm: one forward per layer makes every expert resident, so the decode steps run the sync-free batch-1 path
    for l in range(L):
        eng.forward(l, torch.zeros(1, H, dtype=torch.bfloat16, device=DEV), gates[l])
    eng.sync_copies()

    ws32 = [(w[0].float(), [[t.float() for t in ex] for ex in w[1]], [t.float() for t in w[2]] if w[2] else None) for w in ws]
    tok_ref, tok_gpu = 3, 3
    agree, compared, rels, err_gpu_exact, err_ref_exact = 0, 0, [], 0.0, 0.0
    for step in range(STEPS):
        # ---- oracle arm (and the teacher-forced checks of the GPU layers on the oracle's inputs)
        h = (dec.emb[tok_ref].float() + dec.pos[step].float()).to(torch.bfloat16)[None]
        for l in range(L):
            h, x = dec.pre_moe(h, l)
            ref = _oracle_block(family, x, ws[l], k)
            got = eng.forward(l, x.to(DEV), gates[l])
            r = eng.routing()
            assert torch.equal(torch.from_numpy(r["topk_idx"]).long().sort(-1).values, ref.topk_idx.sort(-1).values), \
                f"{family} step {step} layer {l}: routing differs"
            if family == "mixtral":  # same order too (descending weight)
                assert torch.equal(torch.from_numpy(r["topk_idx"]).long(), ref.topk_idx)
            assert_block_close(got, ref, torch.bfloat16, f"{family} teacher-forced step {step} layer {l}")
            h = (h.float() + ref.out[0].float()).to(torch.bfloat16)
        logits_ref = dec.logits(h)[0]
        # ---- free-running GPU arm: its own token, its own hidden state through all L layers
        hg = (dec.emb[tok_gpu].float() + dec.pos[step].float()).to(torch.bfloat16)[None]
        for l in range(L):
            hg, xg = dec.pre_moe(hg, l)
            og = eng.forward(l, xg.to(DEV), gates[l]).cpu()
            hg = (hg.float() + og.float()).to(torch.bfloat16)
        logits_gpu = dec.logits(hg)[0]
        if tok_gpu == tok_ref:  # same input token: the two chains are comparable
            # "exact" arm: the same chain in fp32 end to end (no bf16 rounding anywhere)
            he = dec.emb[tok_ref].float() + dec.pos[step].float()
            he = he[None]
            for l in range(L):
                n32 = he / he.pow(2).mean(-1, keepdim=True).add(1e-6).sqrt()
                he = he + n32 @ dec.lin[l].float().T
                xe = he / he.pow(2).mean(-1, keepdim=True).add(1e-6).sqrt()
                he = he + _oracle_block(family, xe, ws32[l], k).out[0].float()
            ne = he / he.pow(2).mean(-1, keepdim=True).add(1e-6).sqrt()
            logits_exact = (ne @ dec.lm.float().T)[0]
            rels.append((logits_gpu - logits_ref).abs().mean().item() / logits_ref.abs().mean().item())
            err_gpu_exact += (logits_gpu - logits_exact).abs().mean().item()
            err_ref_exact += (logits_ref - logits_exact).abs().mean().item()
            compared += 1
            agree += int(logits_gpu.argmax().item() == logits_ref.argmax().item())
        tok_ref, tok_gpu = int(logits_ref.argmax()), int(logits_gpu.argmax())
    eng.close()
    rels_sorted = sorted(rels)
    median, worst = rels_sorted[len(rels) // 2], rels_sorted[-1]
    print(f"chained {family}: {compared}/{STEPS} steps compared, argmax agreement {agree}/{compared}, mean-relative logit error vs the oracle: "
          f"median {median:.2e}, worst {worst:.2e}; mean |logit error| vs the fp32 chain: gpu {err_gpu_exact / compared:.3e}, oracle {err_ref_exact / compared:.3e}")
    assert compared >= STEPS // 2, f"the arms diverged after {compared} steps"
    assert median <= 1e-3, f"median mean-relative logit error {median:.2e} > 1e-3 over {L} chained layers"
    assert worst <= 2.0 ** -8, f"a step's mean-relative logit error {worst:.2e} exceeds one bf16 ulp"
    assert err_gpu_exact <= 1.15 * err_ref_exact, f"GPU chain is further from the fp32 chain ({err_gpu_exact / compared:.3e}) than the bf16 oracle chain ({err_ref_exact / compared:.3e})"
    assert agree >= compared - 1, f"argmax agreement {agree}/{compared}"
