"""Serving surface on the real path: the OpenAI server's request batcher in front of a toy decoder
whose MoE layers run on the HIP engine.  Concurrent requests are decoded together — the engine sees ONE forward of B rows
per layer and step instead of B forwards of one row — and every request gets exactly the tokens the same batch gives when
the model is called directly; decoded alone (batch-1 kernels) a request gets the same tokens too."""
from concurrent.futures import ThreadPoolExecutor

import pytest
import torch

from helpers import engine_for, make_weights, register_all

pytestmark = pytest.mark.gpu
fastapi_testclient = pytest.importorskip("fastapi.testclient")
DEV = "cuda:0"
L, H, V, F, E, K = 2, 256, 96, 512, 8, 2
PAD, EOS = 0, 1


class Tok:
    pad_token_id, eos_token_id = PAD, EOS

    def encode(self, text):
        return [2 + (ord(c) % (V - 2)) for c in text]

    def decode(self, ids, skip_special_tokens=True):
        return " ".join(str(int(t)) for t in ids if int(t) > 1 or not skip_special_tokens)

    def apply_chat_template(self, conversation, tokenize=False, add_generation_prompt=True):
        return "".join(m["content"] for m in conversation)


class ToyMoELM:
    """position-wise decoder: h = emb[last real token] + pos[number of real tokens]; L x (h += MoE_l(rmsnorm(h))) on the
    engine; greedy argmax over lm_head.  Rows that emitted EOS keep writing PAD, as HF's generate does."""

    def __init__(self):
        g = torch.Generator().manual_seed(4321)
        self.emb = torch.randn(V, H, generator=g).to(torch.bfloat16).to(DEV)
        self.pos = (torch.randn(64, H, generator=g) * 0.5).to(torch.bfloat16).to(DEV)
        self.lm = (torch.randn(V, H, generator=g) / H ** 0.5).to(torch.bfloat16).to(DEV)
        ws = [make_weights("mixtral", H, F, E, 8800 + l, torch.bfloat16) for l in range(L)]
        self.eng = engine_for("mixtral", H, F, E, K, torch.bfloat16, max_tokens=16, num_layers=L)
        for l in range(L):
            register_all(self.eng, ws[l][1], ws[l][2], layer=l)
        self.gates = [w[0].to(DEV) for w in ws]
        self.forward_rows = []  # rows per engine forward (what the batcher changes)

    @staticmethod
    def _rms(h):
        hf = h.float()
        return (hf / hf.pow(2).mean(-1, keepdim=True).add(1e-6).sqrt()).to(torch.bfloat16)

    def generate(self, input_ids, attention_mask=None, max_new_tokens=8, pad_token_id=PAD, **kw):
        ids = input_ids.to(DEV)
        mask = attention_mask.to(DEV) if attention_mask is not None else torch.ones_like(ids)
        last = ids[:, -1]
        n = mask.sum(-1)
        done = torch.zeros(ids.shape[0], dtype=torch.bool, device=DEV)
        out = [ids]
        for _ in range(max_new_tokens):
            h = (self.emb[last].float() + self.pos[n.clamp(max=63)].float()).to(torch.bfloat16)
            for l in range(L):
                self.forward_rows.append(h.shape[0])
                y = self.eng.forward(l, self._rms(h).contiguous(), self.gates[l])
                h = (h.float() + y.float()).to(torch.bfloat16)
            nxt = (self._rms(h).float() @ self.lm.float().T).argmax(-1)
            nxt = torch.where(done, torch.full_like(nxt, pad_token_id), nxt)
            done |= nxt == EOS
            out.append(nxt[:, None])
            last, n = torch.where(done, last, nxt), n + 1
        return torch.cat(out, dim=1).cpu()


def test_batched_serving_runs_one_engine_forward_per_layer_for_all_waiting_requests():
    from moe_infinity_amd.entrypoints.openai.api_server import create_app

    model, tok = ToyMoELM(), Tok()
    prompts = ["alpha", "be", "gamma delta", "x", "epsilon!", "zz top"]
    n_new = 6
    # the same batch, model called directly: left-padded exactly as the batcher pads
    pid = [tok.encode(p) for p in prompts]
    width = max(len(p) for p in pid)
    ids = torch.full((len(pid), width), PAD, dtype=torch.long)
    mask = torch.zeros_like(ids)
    for r, p in enumerate(pid):
        ids[r, width - len(p):] = torch.tensor(p)
        mask[r, width - len(p):] = 1
    direct = model.generate(ids, attention_mask=mask, max_new_tokens=n_new)[:, width:]

    def cut(row):
        row = [int(t) for t in row]
        return row[: row.index(EOS)] if EOS in row else row

    want = [tok.decode(cut(direct[r])) for r in range(len(prompts))]
    # ... each request decoded alone (T = 1: the self-routing batch-1 kernels)
    alone = []
    for p in pid:
        o = model.generate(torch.tensor([p]), attention_mask=torch.ones(1, len(p), dtype=torch.long), max_new_tokens=n_new)[0, len(p):]
        alone.append(tok.decode(cut(o)))
    same_alone = sum(a == w for a, w in zip(alone, want))
    assert same_alone >= len(prompts) - 1, f"batch-1 kernels and the batched path disagree on the greedy tokens: {alone} vs {want}"

    model.forward_rows.clear()
    app = create_app(model, tok, "toy-moe", max_batch=8, window_ms=300.0, device=None)
    with fastapi_testclient.TestClient(app) as c:
        def ask(p):
            return c.post("/v1/completions", json={"model": "toy-moe", "prompt": p, "max_tokens": n_new, "temperature": 0}).json()
        with ThreadPoolExecutor(len(prompts)) as ex:
            res = list(ex.map(ask, prompts))
        st = dict(app.state.batcher.stats)
    got = [r["choices"][0]["text"] for r in res]
    assert st["largest_batch"] >= 2, f"nothing was batched: {st}"
    if st["batches"] == 1:  # everything arrived inside the window: the batch IS the direct call, bit for bit
        assert got == want
        assert set(model.forward_rows) == {len(prompts)} and len(model.forward_rows) == L * n_new
    else:  # split over a few batches by arrival time: same tokens unless a last-bit difference flips an argmax
        assert sum(g == w for g, w in zip(got, want)) >= len(prompts) - 1, (got, want)
    assert len(model.forward_rows) < len(prompts) * L * n_new, "as many engine forwards as without batching"
    model.eng.close()
