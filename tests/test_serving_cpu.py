"""Serving surface: the OpenAI endpoints of the reference's api_server.py over a request batcher.
A character-level tokenizer and a deterministic stand-in for `generate` (left-padding and the attention mask must be
honoured for a batched request to get the answer it gets alone) — no GPU, no checkpoint."""
import threading
import time
from concurrent.futures import ThreadPoolExecutor

import pytest
import torch

from moe_infinity_amd.entrypoints.openai.api_server import RequestBatcher, create_app, parse_prompt_format

fastapi_testclient = pytest.importorskip("fastapi.testclient")

PAD, EOS = 0, 1


class CharTokenizer:
    pad_token_id, eos_token_id = PAD, EOS

    def encode(self, text):
        return [ord(c) for c in text]

    def decode(self, ids, skip_special_tokens=True):
        return "".join(chr(int(t)) for t in ids if not (skip_special_tokens and int(t) < 32))

    def apply_chat_template(self, conversation, tokenize=False, add_generation_prompt=True):
        return "".join(f"<{m['role']}>{m['content']}\n" for m in conversation) + ("<assistant>" if add_generation_prompt else "")


class EchoModel:
    """generate(): row r continues with chr(65 + (sum of its REAL prompt tokens + i) % 26); a prompt whose token sum is a
    multiple of 7 stops after two tokens (EOS, then padding like HF writes for finished rows)."""

    def __init__(self, delay=0.0):
        self.calls = []
        self.delay = delay
        self.lock = threading.Lock()

    def generate(self, input_ids, attention_mask=None, max_new_tokens=16, pad_token_id=PAD, **kw):
        with self.lock:
            self.calls.append((tuple(input_ids.shape), dict(kw, max_new_tokens=max_new_tokens)))
        time.sleep(self.delay)
        assert attention_mask is not None and attention_mask.shape == input_ids.shape
        rows = []
        for r in range(input_ids.shape[0]):
            real = input_ids[r][attention_mask[r].bool()]
            assert bool((input_ids[r][~attention_mask[r].bool()] == pad_token_id).all()), "padding must be pad tokens on the left"
            s = int(real.sum())
            new = [65 + (s + i) % 26 for i in range(max_new_tokens)]
            if s % 7 == 0:
                new = new[:2] + [EOS] + [pad_token_id] * (max_new_tokens - 3) if max_new_tokens >= 3 else new
            rows.append(torch.tensor(new, dtype=torch.long))
        return torch.cat([input_ids, torch.stack(rows)], dim=1)


def expected(prompt_ids, n):
    s = sum(prompt_ids)
    new = [65 + (s + i) % 26 for i in range(n)]
    if s % 7 == 0 and n >= 3:
        new = new[:2]
    return "".join(chr(t) for t in new)


@pytest.fixture()
def client():
    model = EchoModel()
    app = create_app(model, CharTokenizer(), "toy-moe", max_batch=8, window_ms=1.0)
    with fastapi_testclient.TestClient(app) as c:
        c.model = model
        yield c


def test_prompt_forms_of_the_completions_endpoint():
    assert parse_prompt_format("abc") == (False, ["abc"])
    assert parse_prompt_format(["a", "b"]) == (False, ["a", "b"])
    assert parse_prompt_format([5, 6, 7]) == (True, [[5, 6, 7]])
    assert parse_prompt_format([[5, 6], [7]]) == (True, [[5, 6], [7]])
    for bad in ([], [[]], [1.5], {"a": 1}):
        with pytest.raises(ValueError):
            parse_prompt_format(bad)


def test_health_and_model_list(client):
    assert client.get("/health").status_code == 200
    body = client.get("/v1/models").json()
    assert body["object"] == "list" and [m["id"] for m in body["data"]] == ["toy-moe"]


def test_completion_string_and_token_prompts_with_usage(client):
    tok = CharTokenizer()
    r = client.post("/v1/completions", json={"model": "toy-moe", "prompt": "hello", "max_tokens": 6, "temperature": 0}).json()
    ids = tok.encode("hello")
    assert r["object"] == "text_completion" and r["model"] == "toy-moe"
    assert r["choices"][0]["text"] == expected(ids, 6) and r["choices"][0]["index"] == 0
    assert r["usage"] == {"prompt_tokens": 5, "completion_tokens": len(expected(ids, 6)), "total_tokens": 5 + len(expected(ids, 6))}
    # array of token arrays: one choice per prompt, ragged lengths left-padded into ONE generate call
    n_calls = len(client.model.calls)
    prompts = [[72, 105], [72, 105, 33, 33, 33], [40]]
    r = client.post("/v1/completions", json={"model": "toy-moe", "prompt": prompts, "max_tokens": 4, "temperature": 0}).json()
    assert [c["text"] for c in r["choices"]] == [expected(p, 4) for p in prompts]
    assert [c["index"] for c in r["choices"]] == [0, 1, 2]
    assert len(client.model.calls) == n_calls + 1 and client.model.calls[-1][0] == (3, 5)
    assert r["usage"]["prompt_tokens"] == 8
    # echo
    r = client.post("/v1/completions", json={"model": "toy-moe", "prompt": "ab", "max_tokens": 3, "temperature": 0, "echo": True}).json()
    assert r["choices"][0]["text"] == "ab" + expected(tok.encode("ab"), 3)


def test_finish_reason_and_eos_cut(client):
    # a prompt whose token sum is a multiple of 7 ends after two tokens: "stop", and nothing after the EOS leaks out
    p = [70, 7]  # sum 77
    r = client.post("/v1/completions", json={"model": "toy-moe", "prompt": p, "max_tokens": 8, "temperature": 0}).json()
    assert r["choices"][0]["text"] == expected(p, 8) and len(r["choices"][0]["text"]) == 2 and r["choices"][0]["finish_reason"] == "stop"
    assert r["usage"]["completion_tokens"] == 2
    r = client.post("/v1/completions", json={"model": "toy-moe", "prompt": [70, 8], "max_tokens": 8, "temperature": 0}).json()
    assert r["choices"][0]["finish_reason"] == "length" and r["usage"]["completion_tokens"] == 8


def test_chat_completion_uses_the_chat_template(client):
    msgs = [{"role": "user", "content": "hi"}]
    r = client.post("/v1/chat/completions", json={"model": "toy-moe", "messages": msgs, "max_tokens": 5, "temperature": 0}).json()
    ids = CharTokenizer().encode("<user>hi\n<assistant>")
    assert r["object"] == "chat.completion" and r["choices"][0]["message"] == {"role": "assistant", "content": expected(ids, 5)}
    assert r["usage"]["prompt_tokens"] == len(ids)


def test_sampling_parameters_reach_generate(client):
    client.post("/v1/completions", json={"model": "toy-moe", "prompt": "x", "max_tokens": 2, "temperature": 0})
    assert client.model.calls[-1][1] == {"do_sample": False, "max_new_tokens": 2}
    client.post("/v1/completions", json={"model": "toy-moe", "prompt": "x", "max_tokens": 2, "temperature": 0.5, "top_p": 0.9})
    assert client.model.calls[-1][1] == {"do_sample": True, "temperature": 0.5, "top_p": 0.9, "max_new_tokens": 2}


def test_errors_are_openai_error_objects(client):
    r = client.post("/v1/completions", json={"model": "toy-moe", "prompt": [], "max_tokens": 2})
    assert r.status_code == 400 and r.json()["object"] == "error"
    r = client.post("/v1/completions", json={"model": "toy-moe", "prompt": "x", "stream": True})
    assert r.status_code == 400 and "stream" in r.json()["message"]


def test_concurrent_requests_are_decoded_together_and_each_gets_its_own_answer():
    """The reference serves requests one at a time (a one-element queue around generate).  Here requests that wait together
    and share sampling parameters run as one batch; a request with different parameters is not merged into it."""
    model = EchoModel(delay=0.2)  # requests that arrive while a batch is generating wait together for the next one
    app = create_app(model, CharTokenizer(), "toy-moe", max_batch=8, window_ms=300.0)
    prompts = ["a", "bcd", "efghij", "kl", "mnopqrstu", "v"]
    with fastapi_testclient.TestClient(app) as c:
        def ask(p, n=5, t=0):
            return c.post("/v1/completions", json={"model": "toy-moe", "prompt": p, "max_tokens": n, "temperature": t}).json()
        with ThreadPoolExecutor(len(prompts) + 1) as ex:
            futs = [ex.submit(ask, p) for p in prompts] + [ex.submit(ask, "zz", 3)]  # the last one: another max_tokens
            res = [f.result() for f in futs]
        tok = CharTokenizer()
        for p, r in zip(prompts, res):
            assert r["choices"][0]["text"] == expected(tok.encode(p), 5), p
        assert res[-1]["choices"][0]["text"] == expected(tok.encode("zz"), 3)
        st = app.state.batcher.stats
        assert st["requests"] == len(prompts) + 1
        assert st["largest_batch"] >= 2, f"nothing was batched: {st}"
        assert st["batches"] < st["requests"]
        # the odd one out ran in a call of its own parameters
        assert any(kw["max_new_tokens"] == 3 and shape[0] == 1 for shape, kw in model.calls)


def test_max_batch_one_is_the_references_one_at_a_time_queue():
    model = EchoModel(delay=0.01)
    b = RequestBatcher(model, pad_token_id=PAD, eos_token_id=EOS, max_batch=1, window_ms=50.0)
    futs = [b.submit([65 + i], {"max_new_tokens": 2}) for i in range(4)]
    outs = [f.result(timeout=10) for f in futs]
    b.close()
    assert all(shape == (1, 1) for shape, _ in model.calls) and len(model.calls) == 4
    assert outs == [[65 + (65 + i + k) % 26 for k in range(2)] for i in range(4)]  # none of 65..68 is a multiple of 7


def test_a_failing_generate_fails_its_requests_not_the_worker():
    class Boom:
        def __init__(self):
            self.n = 0

        def generate(self, ids, attention_mask=None, **kw):
            self.n += 1
            if self.n == 1:
                raise RuntimeError("out of memory (pretend)")
            return torch.cat([ids, torch.full((ids.shape[0], kw["max_new_tokens"]), 66, dtype=torch.long)], dim=1)

    b = RequestBatcher(Boom(), pad_token_id=PAD, eos_token_id=EOS, max_batch=4, window_ms=1.0)
    f1 = b.submit([70], {"max_new_tokens": 2})
    with pytest.raises(RuntimeError):
        f1.result(timeout=10)
    assert b.submit([70], {"max_new_tokens": 2}).result(timeout=10) == [66, 66]
    with pytest.raises(ValueError):
        b.submit([], {"max_new_tokens": 2})
    b.close()


def test_encoder_decoder_models_return_the_decoder_sequence_only():
    """Switch-Transformers and NLLB-MoE are encoder-decoder models: HF generate returns decoder_start_token + the new tokens,
    no echo of the prompt (round-3 advice: slicing at the prompt width dropped the first `width` generated tokens).  The
    reference's TokenStreamer skips exactly its first put() (api_server.py:109-131), which is right for both kinds."""
    import types

    class Seq2Seq(EchoModel):
        config = types.SimpleNamespace(is_encoder_decoder=True)

        def generate(self, input_ids, attention_mask=None, max_new_tokens=16, pad_token_id=PAD, **kw):
            full = super().generate(input_ids, attention_mask=attention_mask, max_new_tokens=max_new_tokens, pad_token_id=pad_token_id, **kw)
            start = torch.full((input_ids.shape[0], 1), 2, dtype=torch.long)  # decoder_start_token_id
            return torch.cat([start, full[:, input_ids.shape[1]:]], dim=1)

    tok = CharTokenizer()
    b = RequestBatcher(Seq2Seq(), pad_token_id=PAD, eos_token_id=EOS, max_batch=4, window_ms=50.0)
    try:
        prompts = ["a much longer prompt than the answer", "bc", "\x07"]  # the last one stops early (token sum 7)
        futs = [b.submit(tok.encode(p), {"max_new_tokens": 5}) for p in prompts]
        for p, f in zip(prompts, futs):
            assert "".join(chr(t) for t in f.result(timeout=10)) == expected(tok.encode(p), 5), p
    finally:
        b.close()


def test_rows_are_cut_at_any_of_the_models_eos_ids_and_trailing_padding_is_stripped():
    """generation_config.eos_token_id may be a LIST and the pad id a different token (Llama-3 style): a row that ends on the
    second EOS must be cut there (finish_reason "stop", completion_tokens without the padding)."""
    import types

    EOT, PAD2 = 3, 4

    class M:
        generation_config = types.SimpleNamespace(eos_token_id=[EOS, EOT])

        def generate(self, input_ids, attention_mask=None, max_new_tokens=6, pad_token_id=PAD2, **kw):
            rows = []
            for r in range(input_ids.shape[0]):
                new = [70, 71, EOT] + [pad_token_id] * (max_new_tokens - 3) if r == 0 else [72] * (max_new_tokens - 2) + [pad_token_id] * 2
                rows.append(torch.tensor(new, dtype=torch.long))
            return torch.cat([input_ids, torch.stack(rows)], dim=1)

    b = RequestBatcher(M(), pad_token_id=PAD2, eos_token_id=EOS, max_batch=2, window_ms=100.0)
    try:
        f0, f1 = b.submit([65, 66], {"max_new_tokens": 6}), b.submit([67], {"max_new_tokens": 6})
        assert f0.result(timeout=10) == [70, 71]           # cut at EOT, which is not the tokenizer's eos
        assert f1.result(timeout=10) == [72, 72, 72, 72]   # no EOS: trailing pad tokens are not completion tokens
    finally:
        b.close()
