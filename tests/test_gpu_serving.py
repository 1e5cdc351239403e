"""Serving surface on the real path, with an ORACLE arm.

1. The OpenAI server's request batcher in front of a toy decoder whose MoE layers run on the HIP engine
   (``moeinf_moe_forward``).  Concurrent requests are decoded together — the engine sees ONE forward of B rows per layer and
   step — and every served token must be the token the ORACLE chain (the same decoder with ``oracle.moe_ref.block_mixtral``
   on the CPU for the MoE layers) emits at that step.  Teacher-forced, as tests/test_gpu_chained.py does it: the model
   REPORTS its own arg-max and CONTINUES from the oracle's token, so every step is compared on identical histories and a
   last-bit difference cannot snowball; a step may differ only where the oracle's own top-2 logits are closer than bf16
   noise (counted, at most one).
2. ``create_app`` over a ``MoE``-SHAPED object (the reference's ``entrypoints/big_modeling.py:24-224``: ``MoE(path, config)``,
   ``.engine`` with the two pybind objects, ``.generate`` ending in ``clear_expert_cache_counts``) built over THIS
   repository's ``prefetch_op`` from a synthetic 2-layer checkpoint directory (safetensors file + offload directory): the
   parameters are offloaded / registered / fetched through ``prefetch_handle`` exactly as ``OffloadEngine`` does it
   (the call sequence test_gpu_dropin.py replays), the MoE blocks go through ``expert_dispatcher`` + ``dispatch_local``;
   served tokens against the oracle chain again.

Dense parts (embedding, norm, lm_head, the toy "attention" Linear of test 2) are computed by the same code for both arms."""
import json
import os
import types
from concurrent.futures import ThreadPoolExecutor

import pytest
import torch
import torch.nn.functional as F

from helpers import R, engine_for, make_weights, register_all

pytestmark = pytest.mark.gpu
fastapi_testclient = pytest.importorskip("fastapi.testclient")
DEV = "cuda:0"
L, H, V, FF, E, K = 2, 256, 96, 512, 8, 2
PAD, EOS = 0, 1


class Tok:
    pad_token_id, eos_token_id = PAD, EOS

    def __init__(self, vocab=V):
        self.vocab = vocab

    def encode(self, text):
        return [2 + (ord(c) % (self.vocab - 2)) for c in text]

    def decode(self, ids, skip_special_tokens=True):
        return " ".join(str(int(t)) for t in ids if int(t) > 1 or not skip_special_tokens)

    def apply_chat_template(self, conversation, tokenize=False, add_generation_prompt=True):
        return "".join(m["content"] for m in conversation)


def _rms(h):
    hf = h.float()
    return (hf / hf.pow(2).mean(-1, keepdim=True).add(1e-6).sqrt()).to(torch.bfloat16)


class _GreedyLM:
    """HF-style greedy ``generate`` of a position-wise toy decoder: the state of a row is (last real token, number of real
    tokens).  Subclasses supply ``step(last, n) -> logits [B, V]`` (fp32, CPU).  ``teacher``: {prompt tuple: oracle tokens} —
    a row then reports its own arg-max and continues from the oracle's token.  Rows that emitted EOS keep writing PAD."""

    teacher = None

    def generate(self, input_ids, attention_mask=None, max_new_tokens=8, pad_token_id=PAD, **kw):
        ids = input_ids.cpu()
        mask = attention_mask.cpu() if attention_mask is not None else torch.ones_like(ids)
        keys = [tuple(int(t) for t in ids[r][mask[r].bool()]) for r in range(ids.shape[0])]
        last, n = ids[:, -1].clone(), mask.sum(-1)
        done = torch.zeros(ids.shape[0], dtype=torch.bool)
        out = [ids]
        self.margins = []
        for i in range(max_new_tokens):
            logits = self.step(last, n)
            logits[:, PAD] = float("-inf")  # the pad id is never a generated token (the tokenizer would drop it from the text)
            top2 = logits.topk(2, dim=-1).values
            self.margins.append(((top2[:, 0] - top2[:, 1]) / logits[:, 1:].abs().amax(-1)).tolist())
            nxt = logits.argmax(-1)
            nxt = torch.where(done, torch.full_like(nxt, pad_token_id), nxt)
            out.append(nxt[:, None])
            cont = nxt
            if self.teacher is not None:
                cont = torch.tensor([self.teacher[k][i] if i < len(self.teacher[k]) else pad_token_id for k in keys])
            done |= cont == EOS
            last, n = torch.where(done, last, cont), n + 1
        return torch.cat(out, dim=1)


def _compare_with_oracle(served, oracle_tokens, oracle_margins, what):
    """served / oracle_tokens: {prompt: [tokens]} (cut at EOS).  Equal everywhere, except at most one step whose oracle
    margin (top-1 minus top-2 logit, relative to the largest |logit|) is inside bf16 noise."""
    near_ties = 0
    for p, want in oracle_tokens.items():
        got = served[p]
        if got == want:
            continue
        i = next(j for j in range(max(len(got), len(want))) if j >= len(got) or j >= len(want) or got[j] != want[j])
        assert oracle_margins[p][i] < 2.0 ** -6, f"{what}: prompt {p!r} step {i}: served {got} but the oracle chain says {want} (margin {oracle_margins[p][i]:.3e})"
        near_ties += 1
    assert near_ties <= 1, f"{what}: {near_ties} prompts differ from the oracle chain at a near-tie"
    return near_ties


# ---------------------------------------------------------------------------------------------------------------------
# 1. the fused engine behind the server
# ---------------------------------------------------------------------------------------------------------------------
class ToyMoELM(_GreedyLM):
    """h = emb[last] + pos[n]; L x (h += MoE_l(rmsnorm(h))); logits = rmsnorm(h) @ lm^T.  moe = "engine": the HIP engine;
    "oracle": oracle.moe_ref.block_mixtral on the CPU."""

    def __init__(self, moe):
        g = torch.Generator().manual_seed(4321)
        self.emb = torch.randn(V, H, generator=g).to(torch.bfloat16)
        self.pos = (torch.randn(64, H, generator=g) * 0.5).to(torch.bfloat16)
        self.lm = (torch.randn(V, H, generator=g) / H ** 0.5).to(torch.bfloat16)
        self.ws = [make_weights("mixtral", H, FF, E, 8800 + l, torch.bfloat16) for l in range(L)]
        self.moe = moe
        self.forward_rows = []  # rows per engine forward (what the batcher changes)
        if moe == "engine":
            self.eng = engine_for("mixtral", H, FF, E, K, torch.bfloat16, max_tokens=16, num_layers=L)
            for l in range(L):
                register_all(self.eng, self.ws[l][1], self.ws[l][2], layer=l)
            self.gates = [w[0].to(DEV) for w in self.ws]

    def step(self, last, n):
        h = (self.emb[last].float() + self.pos[n.clamp(max=63)].float()).to(torch.bfloat16)
        for l in range(L):
            x = _rms(h).contiguous()
            if self.moe == "engine":
                self.forward_rows.append(x.shape[0])
                y = self.eng.forward(l, x.to(DEV), self.gates[l]).cpu()
            else:
                y = R.block_mixtral(x[None], self.ws[l][0], self.ws[l][1], top_k=K).out[0]
            h = (h.float() + y.float()).to(torch.bfloat16)
        return _rms(h).float() @ self.lm.float().T


def _oracle_run(lm, tok, prompts, n_new):
    toks, margins = {}, {}
    for p in prompts:
        ids = tok.encode(p)
        o = lm.generate(torch.tensor([ids]), attention_mask=torch.ones(1, len(ids), dtype=torch.long), max_new_tokens=n_new)[0, len(ids):]
        row = [int(t) for t in o]
        toks[p] = row[: row.index(EOS)] if EOS in row else row
        margins[p] = [m[0] for m in lm.margins]
    return toks, margins


def test_served_tokens_equal_the_oracle_chain_and_waiting_requests_share_one_engine_forward():
    from moe_infinity_amd.entrypoints.openai.api_server import create_app

    tok = Tok()
    prompts = ["alpha", "be", "gamma delta", "x", "epsilon!", "zz top"]
    n_new = 6
    want, margins = _oracle_run(ToyMoELM("oracle"), tok, prompts, n_new)
    model = ToyMoELM("engine")
    model.teacher = {tuple(tok.encode(p)): want[p] + [EOS] for p in prompts}  # (a row the oracle ended continues into EOS)
    app = create_app(model, tok, "toy-moe", max_batch=8, window_ms=300.0, device=None)
    with fastapi_testclient.TestClient(app) as c:
        def ask(p):
            return c.post("/v1/completions", json={"model": "toy-moe", "prompt": p, "max_tokens": n_new, "temperature": 0}).json()
        with ThreadPoolExecutor(len(prompts)) as ex:
            res = list(ex.map(ask, prompts))
        st = dict(app.state.batcher.stats)
        # the same requests one at a time: the T = 1 self-routing kernels instead of the batched ones
        rows_batched = list(model.forward_rows)
        model.forward_rows.clear()
        alone = [ask(p) for p in prompts]
    served = {p: [int(t) for t in r["choices"][0]["text"].split()] for p, r in zip(prompts, res)}
    served_alone = {p: [int(t) for t in r["choices"][0]["text"].split()] for p, r in zip(prompts, alone)}
    ties = _compare_with_oracle(served, want, margins, "batched serving")
    ties_alone = _compare_with_oracle(served_alone, want, margins, "one request at a time (batch-1 kernels)")
    print(f"serving vs oracle chain: {len(prompts)} prompts x {n_new} tokens, near-tie differences {ties} (batched) / {ties_alone} (alone); batcher {st}")
    for p, r in zip(prompts, res):  # usage accounting and finish reasons follow the served tokens
        n = len(served[p])
        assert r["usage"]["completion_tokens"] == n and r["choices"][0]["finish_reason"] == ("length" if n >= n_new else "stop")
    assert st["largest_batch"] >= 2, f"nothing was batched: {st}"
    assert len(rows_batched) < len(prompts) * L * n_new, "as many engine forwards as without batching"
    if st["batches"] == 1:
        assert set(rows_batched) == {len(prompts)} and len(rows_batched) == L * n_new
    assert set(model.forward_rows) == {1}
    model.eng.close()


# ---------------------------------------------------------------------------------------------------------------------
# 2. a MoE()-shaped object over prefetch_op behind the server
# ---------------------------------------------------------------------------------------------------------------------
def _write_checkpoint(path):
    """the synthetic checkpoint directory: config.json + model.safetensors (the toy decoder of oracle/gen_offload_trace.py —
    the model whose OffloadEngine call trace is the golden fixture — plus an embedding table)"""
    from safetensors.torch import save_file

    from oracle.gen_offload_trace import SEED, build_model

    class _Expert(torch.nn.Module):
        def __init__(self, cfg):
            super().__init__()
            self.w1 = torch.nn.Linear(cfg.hidden_size, cfg.intermediate_size, bias=False)
            self.w2 = torch.nn.Linear(cfg.intermediate_size, cfg.hidden_size, bias=False)
            self.w3 = torch.nn.Linear(cfg.hidden_size, cfg.intermediate_size, bias=False)

    class _Block(torch.nn.Module):
        def __init__(self, cfg):
            super().__init__()
            self.gate = torch.nn.Linear(cfg.hidden_size, cfg.num_local_experts, bias=False)
            self.experts = torch.nn.ModuleList([_Expert(cfg) for _ in range(cfg.num_local_experts)])

    sd = {n: t.detach().clone().contiguous() for n, t in build_model(_Block).state_dict().items()}
    gold = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "offload_engine_trace.json")))
    sh = gold["shapes"]
    g = torch.Generator().manual_seed(SEED + 7)
    sd["embed_tokens.weight"] = torch.randn(sh["V"], sh["H"], generator=g).to(torch.bfloat16)
    os.makedirs(path, exist_ok=True)
    save_file(sd, os.path.join(path, "model.safetensors"))
    json.dump({"architectures": ["ToyMixtralForCausalLM"], "hidden_size": sh["H"], "intermediate_size": sh["F"], "num_local_experts": sh["E"],
               "num_experts_per_tok": sh["K"], "num_hidden_layers": sh["L"], "vocab_size": sh["V"]}, open(os.path.join(path, "config.json"), "w"))
    return gold


class ToyMoE(_GreedyLM):
    """The reference's ``MoE`` in shape (entrypoints/big_modeling.py:24-224): ``MoE(model_name_or_path, config)`` loads a
    checkpoint directory, builds the engine-side objects the reference's OffloadEngine builds (``engine.archer_engine`` =
    prefetch_handle, ``engine.expert_dispatcher``, ``engine.expert_executor``; model_offload.py:143-145, 471-477, 751-873) and
    exposes HF's ``generate``; every parameter lives in the offload directory and reaches the GPU through begin/end, every MoE
    block through dispatch_local.  moe = "oracle": the same decoder with plain CPU tensors and oracle.moe_ref for the blocks."""

    def __init__(self, model_name_or_path, config, moe="prefetch_op", gold=None):
        from safetensors.torch import load_file

        self.cfg = types.SimpleNamespace(**json.load(open(os.path.join(model_name_or_path, "config.json"))))
        self.config = types.SimpleNamespace(is_encoder_decoder=False)
        sd = load_file(os.path.join(model_name_or_path, "model.safetensors"))
        self.emb = sd.pop("embed_tokens.weight")
        self.moe = moe
        c = self.cfg
        self.Ln, self.En, self.Kn = c.num_hidden_layers, c.num_local_experts, c.num_experts_per_tok
        if moe == "oracle":
            self.sd = sd
            return
        from moe_infinity_amd import prefetch_op as P
        from moe_infinity_amd.expert_executor import DistributedExpertExecutor

        self.P = P
        nid, topo = gold["name_id_map"], gold["topology"]
        assert sorted(nid) == sorted(sd), "the checkpoint is the model the golden topology was recorded for"
        P.configure(dense_cache_fraction=0.7, device_memory_bytes=0, max_tokens=16, top_k=self.Kn)
        handle = P.prefetch_handle(config["offload_path"], config["device_memory_ratio"])
        self.params = {}
        for name, tid in nid.items():  # OffloadEngine._offload_state_dict (:885-906) + apply_to_model_decorator (:183-193)
            handle.offload(sd[name], tid)
            p = torch.nn.Parameter(torch.zeros(1, dtype=sd[name].dtype), requires_grad=False)
            handle.register(p.data, tid)
            self.params[name] = p
        handle.set_topology([(name, groups) for name, groups in topo])                       # setup_archer_hooks (:751-873)
        disp = P.expert_dispatcher(self.En, self.Ln, 0, 4, 8)
        for name, groups in topo:
            if name.endswith("experts"):
                layer = int(name.split(".")[1])
                for e_, ids in enumerate(groups):
                    disp.register_expert(layer, e_, ids)
        ex = DistributedExpertExecutor(None)
        ex.set_expert_dispatcher(disp)
        self.topo = dict((name, groups) for name, groups in topo)
        self.nid = nid
        self.engine = types.SimpleNamespace(archer_engine=handle, expert_dispatcher=disp, expert_executor=ex)

    def _dense(self, name, h):  # a module under the reference's pre/post-forward hooks (:925-979): begin -> use -> end
        if self.moe == "oracle":
            return F.linear(h, self.sd[name + ".weight"], self.sd.get(name + ".bias"))
        hd = self.engine.archer_engine
        w, b = self.params[name + ".weight"], self.params.get(name + ".bias")
        hd.begin(0, w)
        if b is not None:
            hd.begin(0, b)
        assert w.is_cuda  # begin() re-pointed the parameter at the node's device slab
        y = F.linear(h, w.data.cpu(), None if b is None else b.data.cpu())  # the bytes come from the GPU, the arithmetic is the oracle arm's
        hd.end(0, w)
        if b is not None:
            hd.end(0, b)
        return y

    def _block(self, l, hs):
        # SyncMixtralSparseMoeBlock.forward (mixtral.py:42-101); router arithmetic on the CPU for both arms
        logits = self._dense(f"layers.{l}.block_sparse_moe.gate", hs)                        # :46
        rw = torch.softmax(logits, dim=1, dtype=torch.float)
        rw, sel = torch.topk(rw, self.Kn, dim=-1)
        rw = (rw / rw.sum(-1, keepdim=True)).to(hs.dtype)
        one = torch.nn.functional.one_hot(sel, num_classes=self.En)
        wmask = (rw[:, :, None] * one).permute(0, 2, 1).sum(-1)
        rmask = one.permute(0, 2, 1).sum(-1) > 0
        if self.moe == "oracle":
            pre = f"layers.{l}.block_sparse_moe."
            experts = [[self.sd[f"{pre}experts.{e_}.w{j}.weight"] for j in (1, 2, 3)] for e_ in range(self.En)]
            res = R.dispatch_local(hs, rmask, l, experts, R.MIXTRAL_DENSE_ACT_DENSE)         # the oracle's dispatch + expert FFN
        else:
            res = self.engine.expert_executor.dispatch_local(hs.to(DEV), rmask.to(DEV), l)   # :92-94 -> expert_dispatcher on the GPU
        fin = torch.zeros_like(hs)
        for out, _, idx, _ in res:                                                           # :95-100
            t = rmask[:, idx].bool()
            fin[t, :] += out.cpu() * wmask[t, idx][:, None]
        return fin

    def step(self, last, n):
        h = self.emb[last]
        for l in range(self.Ln):
            if self.moe != "oracle":
                self.engine.archer_engine.fetch_tensors(0, self.topo[f"layers.{l}"][0])      # gen_args_hook (:775-783)
            h = h + self._dense(f"layers.{l}.attn", h)
            h = h + self._block(l, h)
        if self.moe != "oracle":
            self.engine.archer_engine.fetch_tensors(0, self.topo["lm_head"][0])
        return self._dense("lm_head", h).float()

    def generate(self, input_ids, **kwargs):
        out = super().generate(input_ids, **kwargs)
        if self.moe != "oracle":
            self.engine.expert_dispatcher.clear_expert_cache_counts()                        # big_modeling.py:195
        return out

    def close(self):
        if self.moe != "oracle":
            self.engine.archer_engine.clean_up_resources()
            self.P.configure(dense_cache_fraction=0.7, device_memory_bytes=0, max_tokens=256, top_k=0)


def test_server_over_a_moe_shaped_object_on_prefetch_op_from_a_checkpoint_directory(tmp_path):
    from moe_infinity_amd.entrypoints.openai.api_server import create_app

    ckpt = str(tmp_path / "toy-mixtral")
    gold = _write_checkpoint(ckpt)
    tok = Tok(vocab=gold["shapes"]["V"])
    prompts = ["hello", "offloaded experts", "q", "mixture", "z9"]
    n_new = 5
    want, margins = _oracle_run(ToyMoE(ckpt, None, moe="oracle"), tok, prompts, n_new)
    model = ToyMoE(ckpt, {"offload_path": str(tmp_path / "offload"), "device_memory_ratio": 0.5}, gold=gold)
    try:
        assert os.path.exists(os.path.join(str(tmp_path / "offload"), "archer_index")) or os.listdir(str(tmp_path / "offload")), "nothing was offloaded"
        model.teacher = {tuple(tok.encode(p)): want[p] + [EOS] for p in prompts}
        app = create_app(model, tok, "toy-mixtral", max_batch=8, window_ms=300.0, device=None)
        with fastapi_testclient.TestClient(app) as c:
            def ask(p):
                return c.post("/v1/chat/completions", json={"model": "toy-mixtral", "messages": [{"role": "user", "content": p}], "max_tokens": n_new, "temperature": 0}).json()
            with ThreadPoolExecutor(len(prompts)) as ex:
                res = list(ex.map(ask, prompts))
            st = dict(app.state.batcher.stats)
        served = {p: [int(t) for t in r["choices"][0]["message"]["content"].split()] for p, r in zip(prompts, res)}
        ties = _compare_with_oracle(served, want, margins, "MoE-shaped object on prefetch_op")
        hr = model.engine.archer_engine.get_hit_rate()
        print(f"MoE-shaped serving vs oracle chain: {len(prompts)} prompts x {n_new} tokens, near-tie differences {ties}; batcher {st}; hit-rate table {tuple(hr.shape)}")
        assert st["requests"] == len(prompts) and st["largest_batch"] >= 2, st
    finally:
        model.close()
