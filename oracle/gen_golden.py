#!/usr/bin/env python3
"""Generate tests/golden/*.npz by running the REFERENCE's own Python code on CPU.

TEST INFRASTRUCTURE ONLY.  Runs in the build container (needs /root/reference);
the fixtures it writes are committed, because /root/reference does not exist on
the GPU box.  Re-run:  python oracle/gen_golden.py

What is real reference code here (imported from /root/reference, never copied):
  * moe_infinity/models/mixtral.py        SyncMixtralSparseMoeBlock.forward
  * moe_infinity/models/deepseek.py       DeepseekMoEBlock.forward
  * moe_infinity/models/modeling_deepseek MoEGate, DeepseekV2MLP
  * moe_infinity/models/switch_transformers.py  SyncSwitchTransformersSparseMLP.forward
  * moe_infinity/models/nllb_moe.py       SyncNllbMoeSparseMLP.forward
  * moe_infinity/distributed/expert_executor.py  DistributedExpertExecutor.dispatch_local
  * moe_infinity/memory/expert_{tracer,predictor,prefetcher}.py
What is a stand-in (the reference's native core cannot be built without CUDA):
  * ``FakeDispatcher`` plays the pybind ``expert_dispatcher`` object
    (core/python/py_archer_prefetch.cpp:84-92): it runs each enqueued expert with the
    block's own HF expert modules on CPU, which is the same math as
    core/parallel/expert_module.cpp, and returns results in enqueue order.
  * ``MixtralBlockSparseTop2MLP`` (removed from transformers 5.x) is re-declared with
    the 4.37 definition so models/mixtral.py imports.
  * the HF 5.15 NLLB router no longer flattens [B,S,H] to [B*S,H] itself (4.37 did); the
    generator flattens before calling it and keeps the 4.37 return arity.
  * the HF 5.15 Switch router returns (probs, index, logits); it is adapted back to the
    4.37 order (index, probs, logits) the reference unpacks, and its keepdim quirk is
    bypassed by calling the 4.37 formula on its own classifier weights.
"""
import importlib
import os
import sys
import types

sys.dont_write_bytecode = True
import numpy as np
import torch
import torch.nn as nn

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")


def _stub_pkg(name, path):
    m = types.ModuleType(name)
    m.__path__ = [path]
    m.__package__ = name
    sys.modules[name] = m
    return m


def import_reference():
    _stub_pkg("moe_infinity", f"{REF}/moe_infinity")
    u = _stub_pkg("moe_infinity.utils", f"{REF}/moe_infinity/utils")
    mem = _stub_pkg("moe_infinity.memory", f"{REF}/moe_infinity/memory")
    _stub_pkg("moe_infinity.models", f"{REF}/moe_infinity/models")
    _stub_pkg("moe_infinity.distributed", f"{REF}/moe_infinity/distributed")
    import transformers.utils.import_utils as iu

    if not hasattr(iu, "is_torch_fx_available"):
        iu.is_torch_fx_available = lambda: False
    hf = importlib.import_module("moe_infinity.utils.hf_config")
    cfg = importlib.import_module("moe_infinity.utils.config")
    u.parse_moe_param = hf.parse_moe_param
    u.ArcherConfig = cfg.ArcherConfig
    for n in ("expert_tracer", "expert_predictor", "expert_prefetcher"):
        importlib.import_module(f"moe_infinity.memory.{n}")
    mem.ExpertPredictor = sys.modules["moe_infinity.memory.expert_predictor"].ExpertPredictor
    mem.ExpertTracer = sys.modules["moe_infinity.memory.expert_tracer"].ExpertTracer

    import transformers.models.mixtral.modeling_mixtral as mm
    from transformers.activations import ACT2FN

    class MixtralBlockSparseTop2MLP(nn.Module):  # transformers 4.37 definition
        def __init__(self, config):
            super().__init__()
            self.w1 = nn.Linear(config.hidden_size, config.intermediate_size, bias=False)
            self.w2 = nn.Linear(config.intermediate_size, config.hidden_size, bias=False)
            self.w3 = nn.Linear(config.hidden_size, config.intermediate_size, bias=False)
            self.act_fn = ACT2FN[config.hidden_act]

        def forward(self, hidden_states):
            return self.w2(self.act_fn(self.w1(hidden_states)) * self.w3(hidden_states))

    mm.MixtralBlockSparseTop2MLP = MixtralBlockSparseTop2MLP
    mods = {}
    for n in ("mixtral", "deepseek", "switch_transformers", "nllb_moe", "grok"):
        mods[n] = importlib.import_module(f"moe_infinity.models.{n}")
    mods["executor"] = importlib.import_module("moe_infinity.distributed.expert_executor")
    return mods


class FakeDispatcher:
    """Stands in for prefetch_op.expert_dispatcher (memory tier removed)."""

    def __init__(self, get_expert):
        self.get_expert = get_expert
        self.queue = []

    def set_inputs(self, hidden, router_mask):
        self.hidden = hidden.clone()
        self.mask = router_mask.clone()

    def set_expected_queue(self, n):
        self.expected = n

    def enqueue_expert(self, layer, expert, gpu, remote):
        self.queue.append((layer, expert))

    def wait_expert(self):
        assert len(self.queue) == self.expected
        res = []
        e_total = self.mask.shape[-1]
        for layer, e in self.queue:
            tok = self.mask.reshape(-1, e_total)[:, e].bool()
            x = self.hidden.reshape(-1, self.hidden.shape[-1])[tok]
            with torch.no_grad():
                y = self.get_expert(e)(x)
            res.append((y.to(self.hidden.dtype), layer, e, 1))
        self.queue = []
        return res


def npf(t):
    t = t.detach()
    if t.dtype in (torch.bfloat16, torch.float16):
        t = t.float()
    if t.dtype == torch.bool:
        t = t.to(torch.uint8)
    return t.cpu().numpy()


sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from synth import acts, checksum, make_weights  # noqa: E402


def make_executor(mods, dispatcher):
    ex = mods["executor"].DistributedExpertExecutor(None)
    ex.set_expert_dispatcher(dispatcher)
    return ex


def gen_mixtral(mods, name, b, s, h, f, e, k, seed, dtype=torch.bfloat16):
    from transformers import MixtralConfig

    cfg = MixtralConfig(hidden_size=h, intermediate_size=f, num_local_experts=e, num_experts_per_tok=k,
                        num_hidden_layers=1, num_attention_heads=4, num_key_value_heads=4, vocab_size=32)
    blk = mods["mixtral"].SyncMixtralSparseMoeBlock(cfg).to(dtype)
    gate, experts, _ = make_weights("mixtral", h, f, e, seed, dtype)
    with torch.no_grad():
        blk.gate.weight.copy_(gate)
        for ex, (w1, w2, w3) in zip(blk.experts, experts):
            ex.w1.weight.copy_(w1)
            ex.w2.weight.copy_(w2)
            ex.w3.weight.copy_(w3)
    blk.layer_id = 0
    blk.expert_executor = make_executor(mods, FakeDispatcher(lambda i: blk.experts[i]))
    x = acts(b * s, h, dtype, 2024 + seed).reshape(b, s, h)
    with torch.no_grad():
        out, logits = blk(x)
        # re-derive the routing the block used (its locals are not returned); same ops as mixtral.py:48-54
        rw = torch.softmax(logits, dim=1, dtype=torch.float)
        rw, sel = torch.topk(rw, k, dim=-1)
        rw = (rw / rw.sum(-1, keepdim=True)).to(dtype)
    np.savez_compressed(os.path.join(OUT, name), x=npf(x), out=npf(out), logits=npf(logits), topk_idx=npf(sel),
                        topk_w=npf(rw), meta=np.array([b, s, h, f, e, k, seed]), wsum=checksum(gate, experts))
    print(name, "ok", out.float().abs().mean().item())


def gen_grok(mods, name, b, s, h, f, e, k, seed, dtype=torch.bfloat16):
    """SyncGrokMoeBlock.forward (moe_infinity/models/grok.py:34-95): its router (no renormalisation), its dispatch_local call and
    its combine loop are the reference's own code; the expert FFN behind the dispatcher is what the reference's CORE runs for this
    architecture — expert type 4 (common/constants.py:33) over the blob in named_parameters order (linear_v, linear_1, linear),
    i.e. MixtralExpert's silu(x W0^T) * (x W2^T) W1^T (core/parallel/expert_module.cpp:147-175).  The predictor / prefetcher the
    block calls every forward are the reference's own classes over a recording handle (their output does not enter the block's
    result)."""
    import torch.nn.functional as Fn

    blk = mods["grok"].SyncGrokMoeBlock(h, f, e, k).to(dtype)
    gate, experts, _ = make_weights("mixtral", h, f, e, seed, dtype)  # (w1 [F,H], w2 [H,F], w3 [F,H]) = blob tensors 0, 1, 2
    with torch.no_grad():
        blk.gate.weight.copy_(gate)
        for ex, (w1, w2, w3) in zip(blk.experts, experts):
            ex.linear_v.weight.copy_(w1)
            ex.linear_1.weight.copy_(w2)
            ex.linear.weight.copy_(w3)
    order = [n for n, _ in blk.experts[0].named_parameters()]
    assert order == ["linear_v.weight", "linear_1.weight", "linear.weight"], order

    def core_type4(i):
        ex = blk.experts[i]
        t = [p for _, p in ex.named_parameters()]  # the blob's tensor order (model_offload.py:645-671)
        return lambda x: Fn.linear(Fn.silu(Fn.linear(x, t[0])) * Fn.linear(x, t[2]), t[1])

    class _Predictor:
        def predict(self, seq_id, expert_index, layer_id):
            return torch.zeros(1, e)

    class _Prefetcher:
        def prefetch_experts(self, layer_id, expert_matrix):
            return None

    blk.layer_id = 0
    blk.seq_id_list = list(range(b))
    blk.expert_predictor, blk.expert_prefetcher = _Predictor(), _Prefetcher()
    blk.expert_executor = make_executor(mods, FakeDispatcher(core_type4))
    x = acts(b * s, h, dtype, 2024 + seed).reshape(b, s, h)
    with torch.no_grad():
        ret = blk(x)
        out, logits = ret[0], ret[1]
        rw = torch.softmax(logits, dim=1, dtype=torch.float)
        rw, sel = torch.topk(rw, k, dim=-1)
        rw = rw.to(dtype)
    np.savez_compressed(os.path.join(OUT, name), x=npf(x), out=npf(out.reshape(b, s, h)), logits=npf(logits), topk_idx=npf(sel),
                        topk_w=npf(rw), meta=np.array([b, s, h, f, e, k, seed]), wsum=checksum(gate, experts))
    print(name, "ok", out.float().abs().mean().item())


def gen_deepseek(mods, name, b, s, h, f, e, k, n_shared, seed, topk_method="greedy", n_group=None, topk_group=None,
                 norm_topk_prob=False, scaling=1.0, dtype=torch.bfloat16):
    from moe_infinity.models.modeling_deepseek.configuration_deepseek import DeepseekV2Config

    cfg = DeepseekV2Config(hidden_size=h, moe_intermediate_size=f, n_routed_experts=e, num_experts_per_tok=k,
                           n_shared_experts=n_shared, topk_method=topk_method, n_group=n_group, topk_group=topk_group,
                           norm_topk_prob=norm_topk_prob, routed_scaling_factor=scaling, num_hidden_layers=1,
                           vocab_size=32, bos_token_id=None, eos_token_id=None)
    cfg.model_type = "deepseek_v2"
    blk = mods["deepseek"].DeepseekMoEBlock(cfg).to(dtype).eval()
    gate, experts, shared = make_weights("deepseek", h, f, e, seed, dtype, n_shared=n_shared or 0)
    with torch.no_grad():
        blk.gate.weight.copy_(gate)
        mlps = list(blk.experts) + ([blk.shared_experts] if n_shared else [])
        for ex, (g_, u_, d_) in zip(mlps, experts + ([shared] if n_shared else [])):
            ex.gate_proj.weight.copy_(g_)
            ex.up_proj.weight.copy_(u_)
            ex.down_proj.weight.copy_(d_)
    blk.layer_id = 0
    blk.expert_executor = make_executor(mods, FakeDispatcher(lambda i: blk.experts[i]))
    x = acts(b * s, h, dtype, 2024 + seed).reshape(b, s, h)
    with torch.no_grad():
        out = blk(x)
        gi, gw, _ = blk.gate(x)
    np.savez_compressed(os.path.join(OUT, name), x=npf(x), out=npf(out), topk_idx=npf(gi), topk_w=npf(gw),
                        meta=np.array([b, s, h, f, e, k, n_shared or 0, seed]),
                        cfg=np.array([topk_method, str(n_group), str(topk_group), str(int(norm_topk_prob)), str(scaling)]),
                        wsum=checksum(gate, experts, shared))
    print(name, "ok", out.float().abs().mean().item())


def gen_deepseek_v3(mods, name, b, s, h, f, e, k, n_shared, seed, n_group, topk_group, norm_topk_prob=True, scaling=2.5, dtype=torch.bfloat16):
    """DeepseekMoEBlock with the V3 gate (deepseek.py:22-26: MoEGate of modeling_deepseek_v3, sigmoid scores + e_score_correction_bias
    + top-2-sum group selection, modeling_deepseek_v3/modeling_deepseek.py:466-528).  The bias is seeded N(0, 0.1^2) fp32 (it is a
    learned parameter; the model dtype cast of the block turns it into bf16 like every parameter: the golden stores what the gate used)."""
    import importlib

    importlib.import_module("moe_infinity.models.modeling_deepseek_v3")
    from moe_infinity.models.modeling_deepseek_v3.configuration_deepseek import DeepseekV3Config

    cfg = DeepseekV3Config(hidden_size=h, moe_intermediate_size=f, n_routed_experts=e, num_experts_per_tok=k, n_shared_experts=n_shared,
                           n_group=n_group, topk_group=topk_group, norm_topk_prob=norm_topk_prob, routed_scaling_factor=scaling,
                           num_hidden_layers=1, vocab_size=32, bos_token_id=None, eos_token_id=None)
    assert cfg.model_type == "deepseek_v3" and cfg.scoring_func == "sigmoid" and cfg.topk_method == "noaux_tc"
    blk = mods["deepseek"].DeepseekMoEBlock(cfg).to(dtype).eval()
    gate, experts, shared = make_weights("deepseek", h, f, e, seed, dtype, n_shared=n_shared or 0)
    g = torch.Generator().manual_seed(9000 + seed)
    bias = (torch.randn(e, generator=g) * 0.1).to(dtype)  # the block's parameters are in the model dtype
    with torch.no_grad():
        blk.gate.weight.copy_(gate)
        blk.gate.e_score_correction_bias.copy_(bias)
        mlps = list(blk.experts) + ([blk.shared_experts] if n_shared else [])
        for ex, (g_, u_, d_) in zip(mlps, experts + ([shared] if n_shared else [])):
            ex.gate_proj.weight.copy_(g_)
            ex.up_proj.weight.copy_(u_)
            ex.down_proj.weight.copy_(d_)
    blk.layer_id = 0
    blk.expert_executor = make_executor(mods, FakeDispatcher(lambda i: blk.experts[i]))
    x = acts(b * s, h, dtype, 2024 + seed).reshape(b, s, h)
    with torch.no_grad():
        out = blk(x)
        gi, gw = blk.gate(x)[:2]
    np.savez_compressed(os.path.join(OUT, name), x=npf(x), out=npf(out), topk_idx=npf(gi), topk_w=npf(gw), e_bias=npf(blk.gate.e_score_correction_bias),
                        meta=np.array([b, s, h, f, e, k, n_shared or 0, seed]),
                        cfg=np.array(["noaux_tc", str(n_group), str(topk_group), str(int(norm_topk_prob)), str(scaling)]),
                        wsum=checksum(gate, experts, shared))
    print(name, "ok", out.float().abs().mean().item())


class _NoCuda:
    """The Switch/NLLB blocks ship router stats to "cuda:0" (switch_transformers.py:110-113,
    nllb_moe.py:106-109); on the CPU-only build box those .to() calls become no-ops."""

    def __enter__(self):
        self.orig = torch.Tensor.to
        orig = self.orig

        def to_nocuda(t, *a, **kw):
            if a and isinstance(a[0], str) and a[0].startswith("cuda"):
                return t
            return orig(t, *a, **kw)

        torch.Tensor.to = to_nocuda

    def __exit__(self, *exc):
        torch.Tensor.to = self.orig


def gen_switch(mods, name, b, s, h, f, e, cap, seed, dtype=torch.float32):
    from transformers import SwitchTransformersConfig

    cfg = SwitchTransformersConfig(d_model=h, d_ff=f, num_experts=e, expert_capacity=cap, num_layers=2,
                                   num_decoder_layers=2, num_heads=4, d_kv=16, vocab_size=32, dropout_rate=0.0)
    blk = mods["switch_transformers"].SyncSwitchTransformersSparseMLP(cfg).to(dtype).eval()
    gate, experts, _ = make_weights("switch", h, f, e, seed, dtype, gate_std=0.5)
    with torch.no_grad():
        blk.router.classifier.weight.copy_(gate)
        for i, (wi, wo) in enumerate(experts):
            ex = blk.experts[f"expert_{i}"]
            ex.wi.weight.copy_(wi)
            ex.wo.weight.copy_(wo)
    hf_router = blk.router

    class Router437(nn.Module):  # 4.37 return order + per-row cumsum, on the HF module's own classifier
        def forward(self, hidden_states):
            in_dtype = hidden_states.dtype
            hs = hidden_states.to(hf_router.dtype)
            logits = hf_router.classifier.to(hf_router.dtype)(hs)
            probs = torch.softmax(logits, dim=-1, dtype=hf_router.dtype).to(in_dtype)
            idx = torch.argmax(probs, dim=-1)
            oh = torch.nn.functional.one_hot(idx, num_classes=hf_router.num_experts)
            prio = torch.cumsum(oh, dim=-2)
            oh = oh * (prio <= hf_router.expert_capacity)
            return oh, torch.max(probs, dim=-1).values.unsqueeze(-1), logits

    blk.router = Router437()
    blk.layer_id = 0
    blk.expert_executor = make_executor(mods, FakeDispatcher(lambda i: blk.experts[f"expert_{i}"]))
    x = acts(b * s, h, dtype, 2024 + seed).reshape(b, s, h)
    with _NoCuda(), torch.no_grad():
        out, (logits, expert_index) = blk.forward(x)
        mask, probs, _ = blk.router(x)
    np.savez_compressed(os.path.join(OUT, name), x=npf(x), out=npf(out), logits=npf(logits), router_mask=npf(mask),
                        router_probs=npf(probs), expert_index=npf(expert_index),
                        meta=np.array([b, s, h, f, e, cap, seed]), wsum=checksum(gate, experts))
    print(name, "ok", out.float().abs().mean().item(), "dropped", int((mask.sum(-1) == 0).sum()))


def gen_nllb(mods, name, b, s, h, f, e, seed, dtype=torch.bfloat16, norm_before=False):
    from transformers import NllbMoeConfig

    cfg = NllbMoeConfig(d_model=h, encoder_ffn_dim=f, decoder_ffn_dim=f, num_experts=e, encoder_layers=2,
                        decoder_layers=2, encoder_attention_heads=4, decoder_attention_heads=4, vocab_size=32,
                        expert_capacity=64, router_dtype="float32", second_expert_policy="all",
                        normalize_router_prob_before_dropping=norm_before, batch_prioritized_routing=False,
                        moe_eval_capacity_token_fraction=1.0, moe_token_dropout=0.2, activation_dropout=0.0)
    blk = mods["nllb_moe"].SyncNllbMoeSparseMLP(cfg, f).to(dtype).eval()
    gate, experts, _ = make_weights("nllb", h, f, e, seed, dtype, gate_std=0.5)
    with torch.no_grad():
        blk.router.classifier.weight.copy_(gate)
        for i, (w1, b1, w2, b2) in enumerate(experts):
            ex = blk.experts[f"expert_{i}"]
            ex.fc1.weight.copy_(w1)
            ex.fc1.bias.copy_(b1)
            ex.fc2.weight.copy_(w2)
            ex.fc2.bias.copy_(b2)
    blk.layer_id = 0

    def nllb_expert(i):
        # core/parallel/expert_module.cpp:88-93 op sequence (matmul, THEN bias add: two roundings in
        # bf16) on the HF module's parameters; HF's fused F.linear(bias) rounds once and differs by 1 ulp.
        ex = blk.experts[f"expert_{i}"]
        return lambda xx: torch.matmul(torch.relu(torch.matmul(xx, ex.fc1.weight.t()) + ex.fc1.bias),
                                       ex.fc2.weight.t()) + ex.fc2.bias

    blk.expert_executor = make_executor(mods, FakeDispatcher(nllb_expert))
    x = acts(b * s, h, dtype, 2024 + seed).reshape(b, s, h)
    orig_router_fwd = blk.router.forward
    # 4.37 semantics: the router flattened [B,S,H] -> [B*S,H] itself and returned (top_1_mask, probs)
    blk.router.forward = lambda hs, pm=None: orig_router_fwd(hs.reshape(-1, hs.shape[-1]), pm)[:2]
    with _NoCuda(), torch.no_grad():
        out, (router_probs, top1) = blk.forward(x)
    np.savez_compressed(os.path.join(OUT, name), x=npf(x), out=npf(out), router_probs=npf(router_probs), top1=npf(top1),
                        meta=np.array([b, s, h, f, e, seed, int(norm_before)]), wsum=checksum(gate, experts))
    print(name, "ok", out.float().abs().mean().item())


def gen_tracer(mods, name, layers, experts, capacity, n_hist, steps, k, seed):
    """ExpertTracer/ExpertPredictor/ExpertPrefetcher on CPU (device strings patched)."""
    tr_mod = sys.modules["moe_infinity.memory.expert_tracer"]
    pr_mod = sys.modules["moe_infinity.memory.expert_predictor"]
    pf_mod = sys.modules["moe_infinity.memory.expert_prefetcher"]
    cfg = types.SimpleNamespace(architectures=["MixtralForCausalLM"], num_hidden_layers=layers, num_local_experts=experts)
    orig_zeros, orig_to = torch.zeros, torch.Tensor.to

    def zeros_cpu(*a, **kw):
        if str(kw.get("device", "")).startswith("cuda"):
            kw["device"] = "cpu"
        return orig_zeros(*a, **kw)

    def to_nocuda(self, *a, **kw):
        if a and isinstance(a[0], str) and a[0].startswith("cuda"):
            return self
        if a and isinstance(a[0], str) and a[0] == "cpu":
            return self.clone()  # on the real device .to("cpu") copies; the predictor then edits its copy in place
        return orig_to(self, *a, **kw)

    torch.zeros, torch.Tensor.to = zeros_cpu, to_nocuda
    try:
        tr_mod.ExpertTracer._instance = None
        tracer = tr_mod.ExpertTracer(capacity, cfg)
        rng = np.random.default_rng(seed)
        hist = np.zeros((capacity, layers, experts), dtype=np.float32)
        for i in range(n_hist):  # historical EAMs: skewed counts
            pref = rng.dirichlet(np.ones(experts) * 0.3, size=layers)
            for l in range(layers):
                hist[i, l] = rng.multinomial(40 * k, pref[l])
        tracer.trace_collection = torch.from_numpy(hist.copy())
        predictor = pr_mod.ExpertPredictor(cfg)
        predictor.add_tracer(tracer)

        class Eng:
            def __init__(self):
                self.calls = []

            def replace_cache_candidates(self, ids):
                self.calls.append(("protect", list(ids)))

            def get_node_default_device(self, ids):
                return 0

            def enqueue_prefetch(self, tid, gpu):
                self.calls.append(("prefetch", tid))

        pf = pf_mod.ExpertPrefetcher(cfg)
        eng = Eng()
        pf.set_archer_engine(eng)
        pf.expert_tensor_map = {(l, e): l * experts + e for l in range(layers) for e in range(experts)}
        seq = tracer.create_entry()
        idx_log, pred_log, order_log, nearest_log = [], [], [], []
        base = hist[rng.integers(0, n_hist)]
        for st in range(steps):
            for l in range(layers):
                p = base[l] + 0.5
                sel = rng.choice(experts, size=k, replace=False, p=p / p.sum())
                before = tracer.collection_access.copy()
                m = predictor.predict(seq, torch.from_numpy(sel[None, :]), l)
                nearest_log.append(int(np.argmax(tracer.collection_access - before)))
                eng.calls = []
                pf.prefetch_experts(l, m)
                order = [c[1] for c in eng.calls if c[0] == "prefetch"]
                idx_log.append(sel)
                pred_log.append(m.copy())
                order_log.append(np.array(order + [-1] * (layers * experts - len(order))))
        np.savez_compressed(os.path.join(OUT, name), hist=hist, sel=np.array(idx_log), pred=np.array(pred_log),
                            order=np.array(order_log), nearest=np.array(nearest_log),
                            eam=tracer.get_entry(seq).matrix, meta=np.array([layers, experts, capacity, n_hist, steps, k, seed]))
        print(name, "ok", len(idx_log), "predict calls")
    finally:
        torch.zeros, torch.Tensor.to = orig_zeros, orig_to


def main():
    os.makedirs(OUT, exist_ok=True)
    torch.manual_seed(0)
    torch.cuda.device_count = lambda: 1  # dispatch_local does expert_id % device_count (expert_executor.py:49-51)
    mods = import_reference()
    gen_mixtral(mods, "mixtral_decode_b1.npz", 1, 1, 256, 512, 8, 2, seed=1)
    gen_mixtral(mods, "mixtral_decode_b4.npz", 4, 1, 256, 512, 8, 2, seed=2)
    gen_mixtral(mods, "mixtral_prefill_t48.npz", 2, 24, 256, 512, 8, 2, seed=3)
    gen_grok(mods, "grok_decode_b1.npz", 1, 1, 256, 512, 8, 2, seed=13)
    gen_grok(mods, "grok_prefill_t40.npz", 2, 20, 256, 512, 8, 2, seed=14)
    gen_deepseek(mods, "deepseek_decode_b1.npz", 1, 1, 256, 176, 64, 6, 2, seed=4)
    gen_deepseek(mods, "deepseek_prefill_t40.npz", 2, 20, 256, 176, 64, 6, 2, seed=5)
    gen_deepseek(mods, "deepseek_group_t16.npz", 1, 16, 256, 176, 64, 6, 2, seed=6, topk_method="group_limited_greedy",
                 n_group=8, topk_group=3, norm_topk_prob=True, scaling=16.0)
    gen_deepseek_v3(mods, "deepseekv3_decode_b1.npz", 1, 1, 256, 176, 64, 6, 1, seed=15, n_group=8, topk_group=4)
    gen_deepseek_v3(mods, "deepseekv3_prefill_t40.npz", 2, 20, 256, 176, 64, 6, 1, seed=16, n_group=8, topk_group=4)
    gen_deepseek_v3(mods, "deepseekv3_e256_t24.npz", 1, 24, 256, 176, 256, 8, 1, seed=17, n_group=8, topk_group=4)
    gen_switch(mods, "switch_decode_b1.npz", 1, 1, 192, 384, 8, 64, seed=7)
    gen_switch(mods, "switch_prefill_cap.npz", 2, 40, 192, 384, 8, 6, seed=8)
    gen_nllb(mods, "nllb_decode_b8.npz", 8, 1, 256, 512, 16, seed=9)
    gen_nllb(mods, "nllb_prefill_f32.npz", 2, 12, 256, 512, 16, seed=10, dtype=torch.float32, norm_before=True)
    gen_tracer(mods, "tracer_l6_e8_full.npz", 6, 8, 24, 24, 4, 2, seed=11)
    gen_tracer(mods, "tracer_l6_e8_partial.npz", 6, 8, 32, 20, 3, 2, seed=12)  # empty slots -> NaN/argmin quirk


if __name__ == "__main__":
    main()
