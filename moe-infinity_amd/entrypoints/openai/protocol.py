"""OpenAI wire format of the two endpoints the reference's server implements
(``moe_infinity/entrypoints/openai/protocol.py``: request fields the handlers read at ``api_server.py:146-283``, response
objects they return).  Only what those handlers use: model list, (chat) completion request / response, usage.  Streaming
chunks, log-probs and model permissions of the reference's file are not served by its handlers either and are left out."""
import time
import uuid
from typing import Dict, List, Literal, Optional, Union

from pydantic import BaseModel, Field


def random_uuid() -> str:
    return uuid.uuid4().hex


def _now() -> int:
    return int(time.time())


class ErrorResponse(BaseModel):
    object: str = "error"
    message: str
    type: str = "invalid_request_error"
    param: Optional[str] = None
    code: int = 400


class ModelCard(BaseModel):
    id: str
    object: str = "model"
    created: int = Field(default_factory=_now)
    owned_by: str = "moe-infinity"
    root: Optional[str] = None


class ModelList(BaseModel):
    object: str = "list"
    data: List[ModelCard] = Field(default_factory=list)


class UsageInfo(BaseModel):
    prompt_tokens: int = 0
    completion_tokens: int = 0
    total_tokens: int = 0


class _Sampling(BaseModel):
    """Sampling fields shared by both request kinds; ``to_hf_params`` (reference: ``protocol.py:90-97,119-129``) turns them into
    ``generate`` keyword arguments.  temperature 0 means greedy, as OpenAI clients expect."""
    temperature: Optional[float] = 1.0
    top_p: Optional[float] = 1.0
    n: Optional[int] = 1
    stop: Optional[Union[str, List[str]]] = Field(default_factory=list)
    stream: Optional[bool] = False
    presence_penalty: Optional[float] = 0.0
    frequency_penalty: Optional[float] = 0.0
    logit_bias: Optional[Dict[str, float]] = None
    user: Optional[str] = None

    def to_hf_params(self, default_max_tokens: int) -> Dict[str, Union[int, float, bool]]:
        max_new = getattr(self, "max_tokens", None)
        p: Dict[str, Union[int, float, bool]] = {"max_new_tokens": int(max_new if max_new is not None else default_max_tokens)}
        t = 1.0 if self.temperature is None else float(self.temperature)
        if t <= 0.0:
            p["do_sample"] = False
        else:
            p["do_sample"] = True
            p["temperature"] = t
            p["top_p"] = 1.0 if self.top_p is None else float(self.top_p)
        return p


class ChatCompletionRequest(_Sampling):
    model: str
    messages: Union[str, List[Dict[str, str]]]
    temperature: Optional[float] = 0.7
    max_tokens: Optional[int] = None


class CompletionRequest(_Sampling):
    model: str
    prompt: Union[List[int], List[List[int]], str, List[str]]  # a string, array of strings, array of tokens, or array of token arrays
    suffix: Optional[str] = None
    max_tokens: Optional[int] = 16
    echo: Optional[bool] = False
    best_of: Optional[int] = None


class CompletionResponseChoice(BaseModel):
    index: int
    text: str
    logprobs: Optional[dict] = None
    finish_reason: Optional[Literal["stop", "length"]] = None


class CompletionResponse(BaseModel):
    id: str = Field(default_factory=lambda: f"cmpl-{random_uuid()}")
    object: str = "text_completion"
    created: int = Field(default_factory=_now)
    model: str
    choices: List[CompletionResponseChoice]
    usage: UsageInfo


class ChatMessage(BaseModel):
    role: str
    content: str


class ChatCompletionResponseChoice(BaseModel):
    index: int
    message: ChatMessage
    finish_reason: Optional[Literal["stop", "length"]] = None


class ChatCompletionResponse(BaseModel):
    id: str = Field(default_factory=lambda: f"chatcmpl-{random_uuid()}")
    object: str = "chat.completion"
    created: int = Field(default_factory=_now)
    model: str
    choices: List[ChatCompletionResponseChoice]
    usage: UsageInfo
