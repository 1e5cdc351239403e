"""OpenAI-compatible server over a ``generate``-capable model (reference:
``moe_infinity/entrypoints/openai/api_server.py``: ``/health`` :133-136, ``/v1/chat/completions`` :146-204,
``/v1/completions`` :207-262, entry point :265-285).

What is different from the reference.  Its handlers take a token from a one-element queue around ``model.generate``
(:165-167, :223-227): requests are served strictly one at a time, batch 1.  A decode step of an offloaded MoE is bound by
streaming the active experts' weights (Mixtral 704 MB per layer and token), so two requests decoded
together cost barely more than one — the expert FFN kernels take ``T`` rows per expert at the price of one weight pass.
``RequestBatcher`` therefore collects the requests that arrive within a short window and share sampling parameters,
left-pads them to one length and runs ONE ``generate`` for the batch; ``max_batch=1`` is the reference's behaviour.

The model is anything with HF's ``generate(input_ids, attention_mask=..., **kwargs)``: the reference's ``MoE`` object
(``moe_infinity/entrypoints/big_modeling.py``) running on this repository's ``prefetch_op``, or a plain HF
model.  Endpoints, request parsing (the four OpenAI prompt forms, :83-106) and response objects follow the reference."""
import argparse
import asyncio
import os
import threading
import time
from concurrent.futures import Future
from typing import Any, Dict, List, Optional, Sequence, Tuple

import fastapi
import torch
from fastapi.responses import JSONResponse, Response

from .protocol import (
    ChatCompletionRequest,
    ChatCompletionResponse,
    ChatCompletionResponseChoice,
    ChatMessage,
    CompletionRequest,
    CompletionResponse,
    CompletionResponseChoice,
    ErrorResponse,
    ModelCard,
    ModelList,
    UsageInfo,
)

TIMEOUT_KEEP_ALIVE = 5  # seconds (reference :57)


def parse_prompt_format(prompt) -> Tuple[bool, list]:
    """OpenAI accepts "a string, array of strings, array of tokens, or array of token arrays" (reference :83-106).
    Returns (prompts_are_token_ids, list of prompts)."""
    if isinstance(prompt, str):
        return False, [prompt]
    if not isinstance(prompt, list) or len(prompt) == 0:
        raise ValueError("please provide at least one prompt")
    first = prompt[0]
    if isinstance(first, str):
        return False, list(prompt)
    if isinstance(first, int):
        return True, [list(prompt)]
    if isinstance(first, list) and len(first) > 0 and isinstance(first[0], int):
        return True, [list(p) for p in prompt]
    raise ValueError("prompt must be a string, array of strings, array of tokens, or array of token arrays")


class _Job:
    __slots__ = ("ids", "key", "kwargs", "future", "t_submit")

    def __init__(self, ids: List[int], kwargs: Dict[str, Any]):
        self.ids = ids
        self.kwargs = kwargs
        self.key = tuple(sorted(kwargs.items()))
        self.future: Future = Future()
        self.t_submit = time.monotonic()


class RequestBatcher:
    """Runs ``model.generate`` on a worker thread, merging requests that wait together.

    A batch is the oldest waiting request plus every later one with the SAME generate arguments (a batch shares its
    sampling parameters and ``max_new_tokens``), up to ``max_batch``, collected for at most ``window_ms`` after the oldest
    arrived.  Prompts are left-padded with ``pad_token_id`` and masked; each request gets back its own new tokens, cut at
    its first ``eos_token_id``.  ``stats`` counts batches and requests (tests, ``/metrics``-style introspection)."""

    def __init__(self, model, pad_token_id: int, eos_token_id: Optional[int] = None, max_batch: int = 8, window_ms: float = 2.0,
                 device: Optional[str] = None):
        self.model = model
        self.pad = int(pad_token_id)
        self.eos = eos_token_id
        # every id that ends a sequence: the tokenizer's EOS plus the model's generation_config.eos_token_id, which may be a
        # LIST (Llama-3 style: several end tokens and a separate pad id)
        self.eos_ids = set() if eos_token_id is None else {int(eos_token_id)}
        g = getattr(getattr(model, "generation_config", None), "eos_token_id", None)
        if g is not None:
            self.eos_ids.update(int(t) for t in (g if isinstance(g, (list, tuple)) else [g]))
        # encoder-decoder models (Switch-Transformers, NLLB-MoE: two of the four supported families) return the DECODER
        # sequence only — decoder_start_token first, no echo of the prompt; decoder-only models echo the (padded) prompt
        self.encoder_decoder = bool(getattr(getattr(model, "config", None), "is_encoder_decoder", False))
        self.max_batch = max(1, int(max_batch))
        self.window = max(0.0, float(window_ms)) / 1e3
        self.device = device
        self.stats = {"batches": 0, "requests": 0, "largest_batch": 0}
        self._cv = threading.Condition()
        self._queue: List[_Job] = []
        self._stop = False
        self._thread = threading.Thread(target=self._run, name="moeinf-batcher", daemon=True)
        self._thread.start()

    def submit(self, input_ids: Sequence[int], gen_kwargs: Dict[str, Any]) -> Future:
        if len(input_ids) == 0:
            raise ValueError("empty prompt")
        job = _Job([int(t) for t in input_ids], dict(gen_kwargs))
        with self._cv:
            if self._stop:
                raise RuntimeError("batcher is closed")
            self._queue.append(job)
            self._cv.notify()
        return job.future

    def close(self):
        with self._cv:
            self._stop = True
            self._cv.notify()
        self._thread.join(timeout=10)

    def _take_batch(self) -> List[_Job]:
        with self._cv:
            while not self._queue and not self._stop:
                self._cv.wait()
            if not self._queue:
                return []
            head = self._queue[0]
            deadline = head.t_submit + self.window
            while not self._stop:  # give later arrivals the rest of the head's window
                same = sum(1 for j in self._queue if j.key == head.key)
                left = deadline - time.monotonic()
                if same >= self.max_batch or left <= 0:
                    break
                self._cv.wait(timeout=left)
            batch = [j for j in self._queue if j.key == head.key][: self.max_batch]
            taken = set(id(j) for j in batch)
            self._queue = [j for j in self._queue if id(j) not in taken]
            return batch

    def _run(self):
        while True:
            batch = self._take_batch()
            if not batch:
                if self._stop:
                    return
                continue
            try:
                outs = self._generate(batch)
                for job, out in zip(batch, outs):
                    job.future.set_result(out)
            except BaseException as e:  # noqa: BLE001 — a failed batch must fail its requests, not the worker
                for job in batch:
                    if not job.future.done():
                        job.future.set_exception(e)

    def _generate(self, batch: List[_Job]) -> List[List[int]]:
        width = max(len(j.ids) for j in batch)
        ids = torch.full((len(batch), width), self.pad, dtype=torch.long)
        mask = torch.zeros((len(batch), width), dtype=torch.long)
        for r, j in enumerate(batch):
            ids[r, width - len(j.ids):] = torch.tensor(j.ids, dtype=torch.long)
            mask[r, width - len(j.ids):] = 1
        if self.device is not None:
            ids, mask = ids.to(self.device), mask.to(self.device)
        kwargs = dict(batch[0].kwargs)
        if "pad_token_id" not in kwargs:
            kwargs["pad_token_id"] = self.pad
        with torch.no_grad():
            out = self.model.generate(ids, attention_mask=mask, **kwargs)
        out = out.sequences if hasattr(out, "sequences") else out
        self.stats["batches"] += 1
        self.stats["requests"] += len(batch)
        self.stats["largest_batch"] = max(self.stats["largest_batch"], len(batch))
        res = []
        for r in range(len(batch)):
            # the reference's TokenStreamer skips exactly the first put() (the prompt / the decoder start token, :109-131)
            new = [int(t) for t in (out[r, 1:] if self.encoder_decoder else out[r, width:]).tolist()]
            ends = [i for i, t in enumerate(new) if t in self.eos_ids]
            if ends:
                new = new[: ends[0]]  # the rest of the row is padding written after this request finished
            else:
                while new and new[-1] == self.pad:
                    new.pop()
            res.append(new)
        return res


def create_app(model, tokenizer, model_name: str, max_batch: int = 8, window_ms: float = 2.0, default_max_tokens: int = 16,
               device: Optional[str] = None) -> fastapi.FastAPI:
    """The reference's application (:64, :133-262) over a ``RequestBatcher``."""
    import contextlib

    @contextlib.asynccontextmanager
    async def lifespan(app_):
        yield
        app_.state.batcher.close()

    app = fastapi.FastAPI(lifespan=lifespan)
    pad = tokenizer.pad_token_id if getattr(tokenizer, "pad_token_id", None) is not None else getattr(tokenizer, "eos_token_id", 0)
    batcher = RequestBatcher(model, pad_token_id=pad if pad is not None else 0, eos_token_id=getattr(tokenizer, "eos_token_id", None),
                             max_batch=max_batch, window_ms=window_ms, device=device)
    app.state.batcher = batcher

    def error(message: str, code: int = 400) -> JSONResponse:
        return JSONResponse(ErrorResponse(message=message, code=code).model_dump(), status_code=code)

    async def generate_all(prompts_ids: List[List[int]], params: Dict[str, Any]) -> List[List[int]]:
        futures = [batcher.submit(ids, params) for ids in prompts_ids]
        return list(await asyncio.gather(*[asyncio.wrap_future(f) for f in futures]))

    def finish_reason(n_new: int, params: Dict[str, Any]) -> str:
        return "length" if n_new >= int(params["max_new_tokens"]) else "stop"

    @app.get("/health")
    async def health() -> Response:
        return Response(status_code=200)

    @app.get("/v1/models")
    async def show_available_models():
        return JSONResponse(ModelList(data=[ModelCard(id=model_name, root=model_name)]).model_dump())

    @app.post("/v1/chat/completions")
    async def chat_completion(request: ChatCompletionRequest):
        if request.stream:
            return error("stream=true is not supported (the reference's handler does not stream either)")
        messages = request.messages if isinstance(request.messages, list) else [{"role": "user", "content": request.messages}]
        try:
            prompt = tokenizer.apply_chat_template(conversation=messages, tokenize=False, add_generation_prompt=True)
            ids = [int(t) for t in tokenizer.encode(prompt)]
            params = request.to_hf_params(default_max_tokens)
            (new,) = await generate_all([ids], params)
        except ValueError as e:
            return error(str(e))
        text = tokenizer.decode(new, skip_special_tokens=True)
        usage = UsageInfo(prompt_tokens=len(ids), completion_tokens=len(new), total_tokens=len(ids) + len(new))
        choice = ChatCompletionResponseChoice(index=0, message=ChatMessage(role="assistant", content=text), finish_reason=finish_reason(len(new), params))
        return ChatCompletionResponse(model=request.model, choices=[choice], usage=usage)

    @app.post("/v1/completions")
    async def completion(request: CompletionRequest):
        if request.stream:
            return error("stream=true is not supported (the reference's handler does not stream either)")
        try:
            is_tokens, prompts = parse_prompt_format(request.prompt)
            prompt_ids = [[int(t) for t in p] if is_tokens else [int(t) for t in tokenizer.encode(p)] for p in prompts]
            params = request.to_hf_params(default_max_tokens)
            news = await generate_all(prompt_ids, params)  # the prompts of ONE request are a batch already
        except ValueError as e:
            return error(str(e))
        choices, n_prompt, n_new = [], 0, 0
        for i, (ids, new) in enumerate(zip(prompt_ids, news)):
            text = tokenizer.decode(new, skip_special_tokens=True)
            if request.echo:
                text = tokenizer.decode(ids, skip_special_tokens=True) + text
            choices.append(CompletionResponseChoice(index=i, text=text, finish_reason=finish_reason(len(new), params)))
            n_prompt += len(ids)
            n_new += len(new)
        return CompletionResponse(model=request.model, choices=choices,
                                  usage=UsageInfo(prompt_tokens=n_prompt, completion_tokens=n_new, total_tokens=n_prompt + n_new))

    return app


def parse_args(argv=None):
    p = argparse.ArgumentParser(description="MoE-Infinity OpenAI-compatible RESTful API server (MI355X engine underneath).")
    p.add_argument("--host", type=str, default=None, help="host name")
    p.add_argument("--port", type=int, default=8000, help="port number")
    p.add_argument("--model", type=str, required=True)
    p.add_argument("--offload-dir", type=str, required=True)
    p.add_argument("--device-memory-ratio", type=float, default=0.75)
    p.add_argument("--max-batch", type=int, default=8, help="requests decoded together (1 = the reference's one-at-a-time queue)")
    p.add_argument("--batch-window-ms", type=float, default=2.0)
    return p.parse_args(argv)


def main(argv=None):
    """Reference :265-285.  The model object is the reference's own ``MoE`` (its Python runs unmodified on this repository's
    ``prefetch_op``) — this entry point needs that package and a checkpoint, neither of which ships
    here; ``create_app`` is what the tests drive."""
    args = parse_args(argv)
    import uvicorn
    from transformers import AutoTokenizer
    try:
        from moe_infinity import MoE
    except ImportError as e:  # pragma: no cover
        raise SystemExit("the reference package `moe_infinity` (with prefetch_op bound to libmoeinf_hip.so, INTEGRATION.md) is needed to build the model: %s" % e)
    tokenizer = AutoTokenizer.from_pretrained(args.model, trust_remote_code=True)
    model = MoE(args.model, {"offload_path": os.path.join(args.offload_dir, args.model), "device_memory_ratio": args.device_memory_ratio})
    app = create_app(model, tokenizer, args.model, max_batch=args.max_batch, window_ms=args.batch_window_ms, device="cuda:0")
    uvicorn.run(app, host=args.host, port=args.port, log_level="info", timeout_keep_alive=TIMEOUT_KEEP_ALIVE)


if __name__ == "__main__":
    main()
