"""Mirrors of the reference's MoE blocks (moe_infinity/models/*.py) on the fused HIP path.

Same class names, constructor arguments (an HF-style config object), ``forward`` signatures and
return values as the blocks they replace, so HF decoder layers call them unchanged:

  SyncMixtralSparseMoeBlock          moe_infinity/models/mixtral.py:18-118      -> (hidden, router_logits)
  DeepseekMoEBlock                   moe_infinity/models/deepseek.py:8-137      -> hidden
  SyncSwitchTransformersSparseMLP    moe_infinity/models/switch_transformers.py:42-113 -> (hidden, (router_logits, expert_index))
  SyncNllbMoeSparseMLP               moe_infinity/models/nllb_moe.py:21-109     -> (hidden, (router_probs, top_1_expert_index))

The router + mask + dispatch_local + combine of each reference ``forward`` is ONE call here
(``engine.forward`` = moeinf_moe_forward).  The blocks own only the gate/classifier weight (a dense
parameter that lives on the GPU, as in the reference); expert weights are registered with the engine
(``register_experts``) and live in its host arena / HBM cache.  ``expert_predictor`` /
``expert_prefetcher`` hooks are the reference's commented-out prefetch calls, revived: when set, the
block feeds this step's routing to the predictor and issues the prefetch list for the next layers.
"""
from typing import Optional, Sequence

import torch
import torch.nn as nn

from . import config as Cf
from .engine import MoEEngine


class _MoeBlockBase(nn.Module):
    layer_id: int = None

    def __init__(self):
        super().__init__()
        self.engine: Optional[MoEEngine] = None
        self.expert_predictor = None
        self.expert_prefetcher = None
        self.seq_id_list = None

    def attach_engine(self, engine: MoEEngine, layer_id: int):
        self.engine, self.layer_id = engine, layer_id

    def register_experts(self, experts: Sequence[Sequence[torch.Tensor]], shared: Optional[Sequence[torch.Tensor]] = None):
        """experts[e] = tensors of expert e in the reference's blob order."""
        ep, rank = self.engine.cfg.ep_size, self.engine.cfg.ep_rank
        for e, ts in enumerate(experts):
            if e % ep == rank:
                self.engine.register_expert(self.layer_id, e, ts)
        if shared is not None:
            self.engine.register_shared(self.layer_id, shared)

    def _gate_weight(self) -> torch.Tensor:
        raise NotImplementedError

    def _run(self, hidden_states: torch.Tensor, batch_rows: int = 1) -> torch.Tensor:
        if self.engine is None:
            raise RuntimeError("attach_engine() first: the block has no CPU/PyTorch fallback")
        out = self.engine.forward(self.layer_id, hidden_states.contiguous(), self._gate_weight(), batch_rows=batch_rows)
        if self.expert_predictor is not None and self.expert_prefetcher is not None and self.seq_id_list:
            # mixtral.py:71-85 (commented out in the reference): predict + prefetch per sequence
            r = self.engine.routing()
            idx = r["topk_idx"].reshape(len(self.seq_id_list), -1, r["topk_idx"].shape[-1])
            for i, seq_id in enumerate(self.seq_id_list):
                matrix = self.expert_predictor.predict(seq_id, idx[i][idx[i] >= 0], self.layer_id)
                self.expert_prefetcher.prefetch_experts(self.layer_id, matrix)
        return out


class SyncMixtralSparseMoeBlock(_MoeBlockBase):
    def __init__(self, config):
        super().__init__()
        self.hidden_dim = config.hidden_size
        self.ffn_dim = config.intermediate_size
        self.num_experts = config.num_local_experts
        self.top_k = config.num_experts_per_tok
        self.gate = nn.Linear(self.hidden_dim, self.num_experts, bias=False)

    def _gate_weight(self):
        return self.gate.weight

    def forward(self, hidden_states: torch.Tensor):
        final = self._run(hidden_states)
        router_logits, _, _ = self.engine.routing_tensors(logits=True)
        return final, router_logits.to(hidden_states.dtype)

    @staticmethod
    def engine_config(config, num_layers, **kw) -> Cf.EngineConfig:
        return Cf.EngineConfig(num_layers=num_layers, num_experts=config.num_local_experts, expert_type=Cf.EXPERT_MIXTRAL,
                               hidden=config.hidden_size, inter=config.intermediate_size, top_k=config.num_experts_per_tok,
                               router_kind=Cf.ROUTER_MIXTRAL, **kw)


class SyncGrokMoeBlock(_MoeBlockBase):
    """moe_infinity/models/grok.py:12-130 (constructor signature (hidden_dim, ffn_dim, num_experts, top_k), :16-18): softmax ->
    top-k WITHOUT renormalisation (:38-45), dispatch, Mixtral's combine (:80-86) — the one block of the reference whose
    predictor / prefetcher calls are live (:60-68; here: _MoeBlockBase._run).  The experts are registered in the reference's blob
    order (named_parameters of MoeMLP: linear_v, linear_1, linear) and run as expert type 4, which is what the reference's core
    does for this architecture (moe_infinity/common/constants.py:33; core/parallel/expert_module.cpp:147-175)."""

    def __init__(self, hidden_dim: int, ffn_dim: int, num_experts: int, top_k: int):
        super().__init__()
        self.hidden_dim, self.ffn_dim, self.num_experts, self.top_k = hidden_dim, ffn_dim, num_experts, top_k
        self.gate = nn.Linear(self.hidden_dim, self.num_experts, bias=False)

    def _gate_weight(self):
        return self.gate.weight

    def forward(self, hidden_states: torch.Tensor):
        final = self._run(hidden_states)
        router_logits, _, _ = self.engine.routing_tensors(logits=True)
        return final, router_logits.to(hidden_states.dtype)

    @staticmethod
    def engine_config(hidden_dim, ffn_dim, num_experts, top_k, num_layers, **kw) -> Cf.EngineConfig:
        return Cf.EngineConfig(num_layers=num_layers, num_experts=num_experts, expert_type=Cf.EXPERT_MIXTRAL, hidden=hidden_dim,
                               inter=ffn_dim, top_k=top_k, router_kind=Cf.ROUTER_SOFTMAX_TOPK, **kw)


class DeepseekMoEBlock(_MoeBlockBase):
    def __init__(self, config):
        super().__init__()
        self.config = config
        self.num_experts_per_tok = config.num_experts_per_tok
        # MoEGate.weight (modeling_deepseek.py:452-455)
        self.gate = nn.Module()
        self.gate.weight = nn.Parameter(torch.empty((config.n_routed_experts, config.hidden_size)))
        if getattr(config, "topk_method", "greedy") == "noaux_tc":  # DeepSeek-V3's MoEGate (modeling_deepseek_v3/modeling_deepseek.py:455-458)
            self.gate.e_score_correction_bias = nn.Parameter(torch.empty((config.n_routed_experts,)))
        self._bias_f32 = None

    def _gate_weight(self):
        return self.gate.weight

    def forward(self, hidden_states):
        b = getattr(self.gate, "e_score_correction_bias", None)
        if b is not None and (self._bias_f32 is None or self._bias_src != b.data_ptr()):
            # the engine reads the bias as fp32 (the gate adds it to fp32 scores; a bf16 parameter is promoted: same values)
            self._bias_f32 = b.detach().to(device=self.engine.device, dtype=torch.float32).contiguous()
            self._bias_src = b.data_ptr()
            self.engine.set_gate_bias(self.layer_id, self._bias_f32)
        return self._run(hidden_states)

    @staticmethod
    def engine_config(config, num_layers, **kw) -> Cf.EngineConfig:
        # DeepSeek-V3's gate (sigmoid scores + e_score_correction_bias + top-2-sum group selection,
        # moe_infinity/models/modeling_deepseek_v3/modeling_deepseek.py:466-528) is a fused router since round 6; any OTHER
        # combination of scoring function and top-k method is refused loudly instead of being routed with V2's softmax rule (such a
        # model runs through the dense-mask path: its own Python gate + prefetch_op.expert_dispatcher / moeinf_dispatch_mask).
        scoring, method = getattr(config, "scoring_func", "softmax"), getattr(config, "topk_method", "greedy")
        if scoring == "sigmoid" and method == "noaux_tc":  # DeepSeek-V3 (round 6: MOEINF_ROUTER_DEEPSEEK_V3; the bias: attach_engine)
            return Cf.EngineConfig(num_layers=num_layers, num_experts=config.n_routed_experts, expert_type=Cf.EXPERT_DEEPSEEK,
                                   hidden=config.hidden_size, inter=config.moe_intermediate_size, top_k=config.num_experts_per_tok,
                                   router_kind=Cf.ROUTER_DEEPSEEK_V3, shared_inter=(config.n_shared_experts or 0) * config.moe_intermediate_size,
                                   norm_topk_prob=bool(config.norm_topk_prob), routed_scaling_factor=float(config.routed_scaling_factor),
                                   n_group=int(config.n_group), topk_group=int(config.topk_group), **kw)
        if scoring != "softmax" or method == "noaux_tc":
            raise NotImplementedError("gate (scoring_func=%r, topk_method=%r) is not a fused router of this engine: keep the model's "
                                      "Python MoEGate and dispatch through prefetch_op.expert_dispatcher (moeinf_dispatch_mask)" % (scoring, method))
        grouped = getattr(config, "topk_method", "greedy") == "group_limited_greedy"
        return Cf.EngineConfig(num_layers=num_layers, num_experts=config.n_routed_experts, expert_type=Cf.EXPERT_DEEPSEEK,
                               hidden=config.hidden_size, inter=config.moe_intermediate_size,
                               top_k=config.num_experts_per_tok, router_kind=Cf.ROUTER_DEEPSEEK,
                               shared_inter=(config.n_shared_experts or 0) * config.moe_intermediate_size,
                               norm_topk_prob=bool(config.norm_topk_prob), routed_scaling_factor=float(config.routed_scaling_factor),
                               n_group=(config.n_group or 0) if grouped else 0, topk_group=(config.topk_group or 0) if grouped else 0, **kw)


class SyncSwitchTransformersSparseMLP(_MoeBlockBase):
    def __init__(self, config):
        super().__init__()
        self.router = nn.Module()
        self.router.classifier = nn.Linear(config.d_model, config.num_experts, bias=False)
        self.num_experts = config.num_experts

    def _gate_weight(self):
        return self.router.classifier.weight

    def forward(self, hidden_states):
        b = hidden_states.shape[0]
        out = self._run(hidden_states, batch_rows=b)
        logits, idx, _ = self.engine.routing_tensors(logits=True, topk=True)
        router_logits = logits.reshape(*hidden_states.shape[:-1], self.num_experts)
        # expert_index = argmax(router_mask) in the reference (switch_transformers.py:112): capacity-dropped tokens have
        # an all-zero mask row -> 0.  The engine reports dropped pairs as -1 (moeinf_copy_routing_dev).
        expert_index = idx.reshape(b, -1).long().clamp_min(0)
        return out, (router_logits, expert_index)

    @staticmethod
    def engine_config(config, num_layers, **kw) -> Cf.EngineConfig:
        return Cf.EngineConfig(num_layers=num_layers, num_experts=config.num_experts, expert_type=Cf.EXPERT_SWITCH,
                               hidden=config.d_model, inter=config.d_ff, top_k=1, router_kind=Cf.ROUTER_SWITCH,
                               expert_capacity=config.expert_capacity, **kw)


class SyncNllbMoeSparseMLP(_MoeBlockBase):
    def __init__(self, config, ffn_dim: int):
        super().__init__()
        self.router = nn.Module()
        self.router.classifier = nn.Linear(config.d_model, config.num_experts, bias=False)
        self.num_experts = config.num_experts
        self.ffn_dim = ffn_dim

    def _gate_weight(self):
        return self.router.classifier.weight

    def forward(self, hidden_states: torch.Tensor, padding_mask: Optional[torch.Tensor] = None):
        if padding_mask is not None:
            raise NotImplementedError("padding_mask routing is not part of the built path")
        out = self._run(hidden_states)
        _, idx, w = self.engine.routing_tensors(logits=False, topk=True)
        t = idx.shape[0]
        router_probs = torch.zeros((t, self.num_experts), dtype=hidden_states.dtype, device=hidden_states.device)
        router_probs.scatter_(1, idx.long(), w.to(hidden_states.dtype))  # combining weights [T,E]
        return out, (router_probs, idx[:, 0].long())

    @staticmethod
    def engine_config(config, ffn_dim, num_layers, **kw) -> Cf.EngineConfig:
        return Cf.EngineConfig(num_layers=num_layers, num_experts=config.num_experts, expert_type=Cf.EXPERT_NLLB,
                               hidden=config.d_model, inter=ffn_dim, top_k=2, router_kind=Cf.ROUTER_NLLB,
                               norm_topk_prob=bool(config.normalize_router_prob_before_dropping), **kw)
