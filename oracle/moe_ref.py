"""CPU restatement of MoE-Infinity's expert-offload hot path (router -> mask ->
dispatch -> per-expert FFN -> combine).

TEST INFRASTRUCTURE ONLY.  Nothing under ``oracle/`` is part of the product:
only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import it, and only as the checker / the timed CPU baseline.  The
product path (``moe-infinity_amd``) never routes through this file.

Parity status: the reference's own tests pin nothing for this path
(SURVEY.md section 8c).  This restatement is pinned instead against outputs of
the reference's *own Python blocks* executed on CPU in the build container
(``oracle/gen_golden.py`` imports ``/root/reference/moe_infinity/models/*.py``,
``memory/*.py`` and the vendored ``MoEGate`` and writes ``tests/golden/*.npz``).
The per-expert FFN lives in the reference's C++ core
(``core/parallel/expert_module.cpp``): restated below as the same ATen op sequence AND
pinned against that very file — it is host-only libtorch code and compiles here
(``oracle/build_ref.py`` -> ``oracle/_ref/libmoeinf_ref.so``; ``tests/test_ref_pin_cpu.py``
and the goldens ``tests/golden/ref_ffn_*.npz`` compare ``expert_ffn`` with the compiled
modules' ``forward`` bit for bit).  The rest of the C++ core needs the CUDA toolkit.

Every function cites the reference lines it follows (paths relative to
``/root/reference``).

Determinism rules the oracle fixes where the reference leaves them
implementation-defined (SURVEY.md section 7 "hard parts"):
  * gate logits: dot products accumulated in fp64 and rounded ONCE to the
    dtype the reference's ``nn.Linear``/``F.linear`` would return;
  * top-k ties: lowest expert index first (stable descending sort);
  * expert completion order (reference: thread-completion order,
    ``core/parallel/expert_dispatcher.cpp:397-434``): ascending expert id, the
    order ``dispatch_local`` enqueues them (``distributed/expert_executor.py:49-54``).
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence, Tuple

import torch
import torch.nn.functional as F

# expert_type ids: core/parallel/expert_module.h:13-18
SWITCH_DENSE_ACT_DENSE = 0
SWITCH_DENSE_GATED_ACT_DENSE = 1
SWITCH_GATED_DENSE_ACT_DENSE = SWITCH_DENSE_GATED_ACT_DENSE
NLLB_DENSE_ACT_DENSE = 2
FSGPT_DENSE_ACT_DENSE = 3
MIXTRAL_DENSE_ACT_DENSE = 4
DEEPSEEK_DENSE_ACT_DENSE = 5

# dtype ids: core/parallel/expert_module.h:20-23
DTYPE_BF16, DTYPE_F32, DTYPE_F16 = 0, 1, 2
TORCH_DTYPE = {DTYPE_BF16: torch.bfloat16, DTYPE_F32: torch.float32, DTYPE_F16: torch.float16}


# --------------------------------------------------------------------------- #
# helpers
# --------------------------------------------------------------------------- #
def gate_logits(x2d: torch.Tensor, wg: torch.Tensor, out_dtype: torch.dtype) -> torch.Tensor:
    """Router GEMM with the oracle's pinned arithmetic: fp64 accumulate, one rounding."""
    return (x2d.double() @ wg.double().t()).to(out_dtype)


def topk_lowest_index(v: torch.Tensor, k: int) -> Tuple[torch.Tensor, torch.Tensor]:
    """torch.topk(sorted=True) with the tie rule pinned to 'lowest index first'."""
    vals, idx = torch.sort(v, dim=-1, descending=True, stable=True)
    return vals[..., :k].contiguous(), idx[..., :k].contiguous()


def argmax_lowest_index(v: torch.Tensor) -> torch.Tensor:
    return topk_lowest_index(v, 1)[1][..., 0]


# --------------------------------------------------------------------------- #
# R1 routers
# --------------------------------------------------------------------------- #
def route_mixtral(x2d: torch.Tensor, wg: torch.Tensor, top_k: int):
    """moe_infinity/models/mixtral.py:46-54.

    gate in model dtype -> softmax in fp32 -> top-k -> renormalise -> cast back.
    Returns (selected_experts [T,K] int64 descending prob, routing_weights [T,K]
    in x dtype, router_logits [T,E] in x dtype).
    """
    router_logits = gate_logits(x2d, wg, x2d.dtype)
    routing_weights = F.softmax(router_logits, dim=1, dtype=torch.float)
    routing_weights, selected = topk_lowest_index(routing_weights, top_k)
    routing_weights = routing_weights / routing_weights.sum(dim=-1, keepdim=True)
    routing_weights = routing_weights.to(x2d.dtype)
    return selected, routing_weights, router_logits


def route_softmax_topk(x2d: torch.Tensor, wg: torch.Tensor, top_k: int):
    """moe_infinity/models/grok.py:38-45 (= arctic.py:39-45): Mixtral's router WITHOUT the renormalisation —
    gate in model dtype -> softmax in fp32 -> top-k -> cast back."""
    router_logits = gate_logits(x2d, wg, x2d.dtype)
    routing_weights = F.softmax(router_logits, dim=1, dtype=torch.float)
    routing_weights, selected = topk_lowest_index(routing_weights, top_k)
    return selected, routing_weights.to(x2d.dtype), router_logits


def route_deepseek(
    x2d: torch.Tensor,
    wg: torch.Tensor,
    top_k: int,
    topk_method: str = "greedy",
    n_group: int = 1,
    topk_group: int = 1,
    norm_topk_prob: bool = False,
    routed_scaling_factor: float = 1.0,
):
    """moe_infinity/models/modeling_deepseek/modeling_deepseek.py:463-512 (MoEGate.forward).

    Everything in fp32.  The reference uses ``sorted=False`` so only the per-token
    SET (idx -> weight) is defined; this returns descending order.
    """
    n = x2d.shape[0]
    logits = gate_logits(x2d.float(), wg.float(), torch.float32)
    scores = logits.softmax(dim=-1, dtype=torch.float32)
    if topk_method == "greedy":
        topk_weight, topk_idx = topk_lowest_index(scores, top_k)
    elif topk_method == "group_limited_greedy":
        group_scores = scores.view(n, n_group, -1).max(dim=-1).values
        group_idx = topk_lowest_index(group_scores, topk_group)[1]
        group_mask = torch.zeros_like(group_scores)
        group_mask.scatter_(1, group_idx, 1)
        score_mask = group_mask.unsqueeze(-1).expand(n, n_group, scores.shape[1] // n_group).reshape(n, -1)
        tmp_scores = scores.masked_fill(~score_mask.bool(), 0.0)
        topk_weight, topk_idx = topk_lowest_index(tmp_scores, top_k)
    else:
        raise NotImplementedError(topk_method)
    if top_k > 1 and norm_topk_prob:
        denominator = topk_weight.sum(dim=-1, keepdim=True) + 1e-20
        topk_weight = topk_weight / denominator
    else:
        topk_weight = topk_weight * routed_scaling_factor
    return topk_idx, topk_weight, logits


def route_deepseek_v3(x2d: torch.Tensor, wg: torch.Tensor, top_k: int, e_bias: torch.Tensor, n_group: int, topk_group: int,
                      norm_topk_prob: bool = True, routed_scaling_factor: float = 1.0):
    """moe_infinity/models/modeling_deepseek_v3/modeling_deepseek.py:466-528 (MoEGate.forward, scoring_func "sigmoid",
    topk_method "noaux_tc").  Everything in fp32: scores = sigmoid(logits); scores_for_choice = scores +
    e_score_correction_bias; a group's score = the sum of its two best scores_for_choice; the topk_group best groups stay,
    the others' candidates are filled with 0.0; top-k over that; the WEIGHTS are the un-biased scores of the chosen experts,
    normalised (sum + 1e-20) if norm_topk_prob and ALWAYS multiplied by routed_scaling_factor.  ``sorted=False`` in the
    reference: only the per-token set (idx -> weight) is defined; this returns descending order of scores_for_choice."""
    n = x2d.shape[0]
    logits = gate_logits(x2d.float(), wg.float(), torch.float32)
    scores = logits.sigmoid()
    sfc = scores + e_bias.float().unsqueeze(0)
    group_scores = sfc.view(n, n_group, -1).topk(2, dim=-1)[0].sum(dim=-1)
    group_idx = topk_lowest_index(group_scores, topk_group)[1]
    group_mask = torch.zeros_like(group_scores)
    group_mask.scatter_(1, group_idx, 1)
    score_mask = group_mask.unsqueeze(-1).expand(n, n_group, scores.shape[1] // n_group).reshape(n, -1)
    tmp_scores = sfc.masked_fill(~score_mask.bool(), 0.0)
    _, topk_idx = topk_lowest_index(tmp_scores, top_k)
    topk_weight = scores.gather(1, topk_idx)
    if top_k > 1 and norm_topk_prob:
        topk_weight = topk_weight / (topk_weight.sum(dim=-1, keepdim=True) + 1e-20)
    topk_weight = topk_weight * routed_scaling_factor
    return topk_idx, topk_weight, logits


def route_switch(x3d: torch.Tensor, wg: torch.Tensor, expert_capacity: int, router_dtype=torch.float32):
    """HF ``SwitchTransformersTop1Router`` (transformers 4.37-era semantics the reference
    expects: returns ``(expert_index one-hot, router_probs, router_logits)``), called at
    moe_infinity/models/switch_transformers.py:76.

    Third-party arithmetic (``transformers>=4.37.1``, requirements.txt:19), restated:
      probs = softmax(classifier(x.to(router_dtype)), dtype=router_dtype).to(input dtype)
      expert_index = one_hot(argmax(probs)); token_priority = cumsum over the SEQUENCE dim
      (per batch row); tokens with priority > expert_capacity are dropped (mask row -> 0);
      router_probs = max prob, shape [B,S,1].
    """
    b, s, h = x3d.shape
    in_dtype = x3d.dtype
    logits = gate_logits(x3d.reshape(-1, h).to(router_dtype), wg.to(router_dtype), router_dtype).view(b, s, -1)
    probs = F.softmax(logits, dim=-1, dtype=router_dtype).to(in_dtype)
    expert_index = argmax_lowest_index(probs)
    one_hot = F.one_hot(expert_index, num_classes=wg.shape[0])
    token_priority = torch.cumsum(one_hot, dim=-2)
    capacity_mask = token_priority <= expert_capacity
    router_mask = one_hot * capacity_mask
    router_probs = torch.max(probs, dim=-1).values.unsqueeze(-1)
    return router_mask, router_probs, logits


def route_nllb(
    x3d: torch.Tensor,
    wg: torch.Tensor,
    router_dtype=torch.float32,
    normalize_router_prob_before_dropping: bool = False,
    moe_eval_capacity_token_fraction: float = 1.0,
):
    """HF ``NllbMoeTop2Router.forward/route_tokens`` (eval mode, second_expert_policy='all',
    batch_prioritized_routing=False, no padding mask), called at
    moe_infinity/models/nllb_moe.py:53.  Returns (top_1_mask [T,E] int64,
    router_probs [T,E] combining weights in input dtype, router_logits [T,E]).

    Quirks kept: probabilities are cast to the INPUT dtype before top-1 argmax and before
    normalisation (so for bf16 inputs top-1 is chosen among bf16-rounded probabilities),
    while top-2 is the argmax of the fp32 LOGITS with top-1 masked out.
    """
    b, s, h = x3d.shape
    in_dtype = x3d.dtype
    e = wg.shape[0]
    logits = gate_logits(x3d.reshape(-1, h).to(router_dtype), wg.to(router_dtype), router_dtype)
    nb_tokens = logits.shape[0]
    probs = F.softmax(logits, dim=-1, dtype=router_dtype).to(in_dtype)
    top1 = argmax_lowest_index(probs)
    top_1_mask = F.one_hot(top1, num_classes=e)
    logits_except_top_1 = logits.masked_fill(top_1_mask.bool(), float("-inf"))
    top2 = argmax_lowest_index(logits_except_top_1)
    top_2_mask = F.one_hot(top2, num_classes=e)

    def normalize(p, m1, m2):
        p1 = (p * m1).sum(dim=1)
        p2 = (p * m2).sum(dim=1)
        denom = torch.clamp(p1 + p2, min=torch.finfo(p.dtype).eps)
        return p1 / denom, p2 / denom

    if normalize_router_prob_before_dropping:
        p1, p2 = normalize(probs, top_1_mask, top_2_mask)
    locations1 = torch.cumsum(top_1_mask, dim=0) - 1
    locations2 = torch.cumsum(top_2_mask, dim=0) - 1
    locations2 = locations2 + torch.sum(top_1_mask, dim=0, keepdim=True)
    if moe_eval_capacity_token_fraction > 0:
        capacity = math.ceil(moe_eval_capacity_token_fraction * nb_tokens)
    else:
        capacity = 2 * math.ceil(nb_tokens / e)
    top_1_mask = top_1_mask * torch.lt(locations1, capacity)
    top_2_mask = top_2_mask * torch.lt(locations2, capacity)
    if not normalize_router_prob_before_dropping:
        p1, p2 = normalize(probs, top_1_mask, top_2_mask)
    gates1 = p1[:, None] * top_1_mask
    gates2 = p2[:, None] * top_2_mask
    return top_1_mask, gates1 + gates2, logits


# --------------------------------------------------------------------------- #
# R2 mask build
# --------------------------------------------------------------------------- #
def masks_from_topk(selected: torch.Tensor, weights: torch.Tensor, num_experts: int):
    """moe_infinity/models/mixtral.py:56-65 and models/deepseek.py:77-91 (generalised to K):
    router_mask[T,E] bool, routing_weights_mask[T,E] (weight dtype)."""
    one_hot = F.one_hot(selected, num_classes=num_experts)  # [T,K,E]
    routing_weights_mask = (weights[:, :, None] * one_hot).permute(0, 2, 1).sum(dim=-1)
    router_mask = one_hot.permute(0, 2, 1).bool().any(dim=-1)
    return router_mask, routing_weights_mask


# --------------------------------------------------------------------------- #
# R3/R4 dispatch order
# --------------------------------------------------------------------------- #
def dispatch_index(router_mask2d: torch.Tensor):
    """What ``dispatch_local`` + ``GPUFetchFunc`` compute, as index arrays.

    moe_infinity/distributed/expert_executor.py:34-43 (tokens per expert, ascending list of
    active experts) and core/parallel/expert_dispatcher.cpp:274-284 (boolean-mask gather =
    ascending token order inside an expert).

    Returns counts[E], offsets[E+1] (exclusive scan), slot_token[n_pairs] (token id of every
    expert-sorted row), and expert_list (ascending active experts)."""
    mask = router_mask2d.bool()
    counts = mask.sum(dim=0).to(torch.int64)
    offsets = torch.zeros(mask.shape[1] + 1, dtype=torch.int64)
    offsets[1:] = torch.cumsum(counts, 0)
    slot_token = torch.cat([torch.nonzero(mask[:, e]).flatten() for e in range(mask.shape[1])]) if mask.numel() else torch.zeros(0, dtype=torch.int64)
    expert_list = [e for e in range(mask.shape[1]) if counts[e] > 0]
    return counts, offsets, slot_token.to(torch.int64), expert_list


# --------------------------------------------------------------------------- #
# R6 expert FFN
# --------------------------------------------------------------------------- #
def expert_ffn(x: torch.Tensor, tensors: Sequence[torch.Tensor], expert_type: int) -> torch.Tensor:
    """core/parallel/expert_module.cpp — the same ATen op sequence, in the tensors' dtype.

    ``tensors`` is in the reference's blob order:
      mixtral  [w1(F,H), w2(H,F), w3(F,H)]            expert_module.cpp:139-175
      deepseek [gate_proj, up_proj, down_proj]        expert_module.cpp:185-204
      nllb     [fc1.w, fc1.b, fc2.w, fc2.b]           expert_module.cpp:70-93
      switch   [wi(F,H), wo(H,F)]                     expert_module.cpp:17-36
    """
    if expert_type == MIXTRAL_DENSE_ACT_DENSE:
        w1, w2, w3 = tensors
        return torch.matmul(F.silu(torch.matmul(x, w1.t())) * torch.matmul(x, w3.t()), w2.t())
    if expert_type == DEEPSEEK_DENSE_ACT_DENSE:
        g, u, d = tensors
        return torch.matmul(F.silu(torch.matmul(x, g.t())) * torch.matmul(x, u.t()), d.t())
    if expert_type in (NLLB_DENSE_ACT_DENSE, FSGPT_DENSE_ACT_DENSE):
        fc1, b1, fc2, b2 = tensors
        if expert_type == FSGPT_DENSE_ACT_DENSE and x.dtype != fc1.dtype:
            x = x.to(fc1.dtype)
        return torch.matmul(torch.relu(torch.matmul(x, fc1.t()) + b1), fc2.t()) + b2
    if expert_type == SWITCH_GATED_DENSE_ACT_DENSE:  # expert_module.cpp:54-59 (torch::gelu default = the erf form)
        wi_0, wi_1, wo = tensors
        return torch.matmul(F.gelu(torch.matmul(x, wi_0.t())) * torch.matmul(x, wi_1.t()), wo.t())
    if expert_type == SWITCH_DENSE_ACT_DENSE:
        wi, wo = tensors
        return torch.matmul(torch.relu(torch.matmul(x, wi.t().to(x.dtype))), wo.t().to(x.dtype))
    raise NotImplementedError(f"expert_type {expert_type}")


def dispatch_local(hidden2d: torch.Tensor, router_mask2d: torch.Tensor, layer_id: int, experts, expert_type: int):
    """distributed/expert_executor.py:32-58 + core/parallel/expert_dispatcher.cpp:191-434 with the
    memory tier removed: list of (output[t_e,H], layer, expert, hit) in ascending expert order."""
    _, _, _, expert_list = dispatch_index(router_mask2d)
    out = []
    for e in expert_list:
        tok = router_mask2d[:, e].bool()
        y = expert_ffn(hidden2d[tok], experts[e], expert_type).to(hidden2d.dtype)
        out.append((y, layer_id, e, 1))
    return out


# --------------------------------------------------------------------------- #
# whole blocks (R1+R2+R3+R4+R6+R7)
# --------------------------------------------------------------------------- #
@dataclass
class BlockResult:
    out: torch.Tensor
    topk_idx: Optional[torch.Tensor] = None  # [T,K] int64 (descending weight)
    topk_w: Optional[torch.Tensor] = None  # [T,K]
    router_mask: Optional[torch.Tensor] = None  # [T,E] bool
    weights_mask: Optional[torch.Tensor] = None  # [T,E]
    logits: Optional[torch.Tensor] = None
    expert_out: Dict[int, torch.Tensor] = field(default_factory=dict)
    extra: Dict[str, torch.Tensor] = field(default_factory=dict)


def block_mixtral(x3d, wg, experts, top_k=2, layer_id=0) -> BlockResult:
    """moe_infinity/models/mixtral.py:40-118 (SyncMixtralSparseMoeBlock.forward)."""
    b, s, h = x3d.shape
    x = x3d.reshape(-1, h)
    sel, w, logits = route_mixtral(x, wg, top_k)
    router_mask, weights_mask = masks_from_topk(sel, w, wg.shape[0])
    final = torch.zeros((b * s, h), dtype=x.dtype)
    res = dispatch_local(x, router_mask, layer_id, experts, MIXTRAL_DENSE_ACT_DENSE)
    r = BlockResult(out=None, topk_idx=sel, topk_w=w, router_mask=router_mask, weights_mask=weights_mask, logits=logits)
    for output, _, idx, _ in res:
        tok = router_mask[:, idx].bool()
        final[tok, :] += output * weights_mask[tok, idx][:, None]
        r.expert_out[idx] = output
    r.out = final.reshape(b, s, h)
    return r


def block_grok(x3d, wg, experts, top_k=2, layer_id=0) -> BlockResult:
    """moe_infinity/models/grok.py:34-95 (SyncGrokMoeBlock.forward): softmax -> top-k, NO renormalisation, the combine of the
    Mixtral block.  ``experts`` = every expert's tensors in the reference's blob order (named_parameters of MoeMLP: linear_v,
    linear_1, linear), run by the reference's core as expert type 4 (moe_infinity/common/constants.py:33: "grok": 4 ->
    MixtralExpert, core/parallel/expert_module.cpp:147-175: silu(x W[0]^T) * (x W[2]^T) then W[1]) — what the reference
    computes for this architecture, which is not the model's own gelu(linear x) * linear_v x."""
    b, s, h = x3d.shape
    x = x3d.reshape(-1, h)
    sel, w, logits = route_softmax_topk(x, wg, top_k)
    router_mask, weights_mask = masks_from_topk(sel, w, wg.shape[0])
    final = torch.zeros((b * s, h), dtype=x.dtype)
    res = dispatch_local(x, router_mask, layer_id, experts, MIXTRAL_DENSE_ACT_DENSE)
    r = BlockResult(out=None, topk_idx=sel, topk_w=w, router_mask=router_mask, weights_mask=weights_mask, logits=logits)
    for output, _, idx, _ in res:
        tok = router_mask[:, idx].bool()
        final[tok, :] += output * weights_mask[tok, idx][:, None]
        r.expert_out[idx] = output
    r.out = final.reshape(b, s, h)
    return r


def block_deepseek(x3d, wg, experts, top_k, shared=None, layer_id=0, **gate_kw) -> BlockResult:
    """moe_infinity/models/deepseek.py:51-137 (DeepseekMoEBlock.forward).  ``shared`` is the
    always-resident shared expert [gate_proj, up_proj, down_proj] (deepseek.py:133-136)."""
    b, s, h = x3d.shape
    x = x3d.reshape(-1, h)
    if "e_bias" in gate_kw:  # DeepSeek-V3's gate (deepseek.py:22-26 picks MoEGate of modeling_deepseek_v3); the block is the same
        idx, w, logits = route_deepseek_v3(x, wg, top_k, **gate_kw)
    else:
        idx, w, logits = route_deepseek(x, wg, top_k, **gate_kw)
    router_mask, weights_mask = masks_from_topk(idx, w, wg.shape[0])
    final = torch.zeros((b * s, h), dtype=x.dtype)
    res = dispatch_local(x, router_mask, layer_id, experts, DEEPSEEK_DENSE_ACT_DENSE)
    r = BlockResult(out=None, topk_idx=idx, topk_w=w, router_mask=router_mask, weights_mask=weights_mask, logits=logits)
    for output, _, e, _ in res:
        tok = router_mask[:, e].bool()
        final[tok, :] += output * weights_mask[tok, e][:, None]
        r.expert_out[e] = output
    final = final.view(b, s, h)
    if shared is not None:
        sh = expert_ffn(x3d, shared, DEEPSEEK_DENSE_ACT_DENSE)
        r.extra["shared_out"] = sh
        final = final + sh
    r.out = final
    return r


def block_switch(x3d, wg, experts, expert_capacity=64, layer_id=0, expert_type=SWITCH_DENSE_ACT_DENSE) -> BlockResult:
    """moe_infinity/models/switch_transformers.py:74-113.  expert_type SWITCH_DENSE_GATED_ACT_DENSE: the same block over the
    gelu-gated experts (T5-v1.1-style Switch checkpoints; moe_infinity/common/constants.py maps them to type 1)."""
    router_mask, router_probs, logits = route_switch(x3d, wg, expert_capacity)
    b, s, h = x3d.shape
    next_states = x3d.clone()
    x = x3d.reshape(-1, h)
    mask2d = router_mask.reshape(-1, router_mask.shape[-1])
    res = dispatch_local(x, mask2d, layer_id, experts, expert_type)
    r = BlockResult(out=None, router_mask=mask2d.bool(), logits=logits.reshape(-1, logits.shape[-1]))
    for output, _, e, _ in res:
        tok = router_mask[:, :, e].bool()
        next_states[tok] = output
        r.expert_out[e] = output
    r.extra["router_probs"] = router_probs
    r.topk_idx = torch.argmax(router_mask, dim=-1).reshape(-1, 1)
    r.out = router_probs * next_states
    return r


def block_nllb(x3d, wg, experts, layer_id=0, **router_kw) -> BlockResult:
    """moe_infinity/models/nllb_moe.py:46-109 (incl. the ``next_states == 0`` passthrough quirk :103)."""
    b, s, h = x3d.shape
    top_1_mask, router_probs, logits = route_nllb(x3d, wg, **router_kw)
    e = wg.shape[0]
    combining = router_probs.reshape(b, s, e)
    router_mask = combining.bool()
    next_states = torch.zeros_like(x3d)
    x = x3d.reshape(-1, h)
    res = dispatch_local(x, router_mask.reshape(-1, e), layer_id, experts, NLLB_DENSE_ACT_DENSE)
    r = BlockResult(out=None, router_mask=router_mask.reshape(-1, e), weights_mask=router_probs, logits=logits)
    for output, _, idx, _ in res:
        tok = router_mask[..., idx].bool()
        wts = combining[..., idx]
        next_states[tok] += torch.einsum("b,be->be", wts[tok], output)
        r.expert_out[idx] = output
    # kept for the parity bar: the `== 0` passthrough below is a discontinuity (oracle/parity.py)
    r.extra["pre_passthrough"] = next_states.clone()
    r.extra["x"] = x3d
    zero = next_states == 0
    next_states[zero] = x3d[zero]
    r.extra["top_1_mask"] = top_1_mask
    r.out = next_states
    return r


# --------------------------------------------------------------------------- #
# synthetic models shared by tests and bench
# --------------------------------------------------------------------------- #
def synth_expert(expert_type: int, h: int, f: int, dtype: torch.dtype, gen: torch.Generator, std=0.02):
    def w(*shape):
        return (torch.randn(*shape, generator=gen, dtype=torch.float32) * std).to(dtype)

    if expert_type == MIXTRAL_DENSE_ACT_DENSE:
        return [w(f, h), w(h, f), w(f, h)]
    if expert_type == DEEPSEEK_DENSE_ACT_DENSE:
        return [w(f, h), w(f, h), w(h, f)]
    if expert_type in (NLLB_DENSE_ACT_DENSE, FSGPT_DENSE_ACT_DENSE):
        return [w(f, h), w(f), w(h, f), w(h)]
    if expert_type == SWITCH_DENSE_ACT_DENSE:
        return [w(f, h), w(h, f)]
    raise NotImplementedError


def synth_activations(t: int, h: int, dtype: torch.dtype, seed: int):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(t, h, generator=g, dtype=torch.float32)
    x = x / x.pow(2).mean(dim=-1, keepdim=True).sqrt()
    return x.to(dtype)
