"""Configuration objects.

``ArcherConfig`` mirrors moe_infinity/utils/config.py:14-77 (same keys, same defaults) so user
configs of the reference load unchanged; ``device_memory_bytes`` and ``cache_policy`` are the two
additions.  ``EngineConfig`` is the typed view handed to the C ABI.
"""
import json
import os
from dataclasses import asdict, dataclass, field, fields
from typing import Optional

# ids shared with include/moeinf.h (and core/parallel/expert_module.h in the reference)
DTYPE_BF16, DTYPE_F32, DTYPE_F16 = 0, 1, 2
DTYPE_F8E4M3 = 3  # fp8 (e4m3fn) experts in the host tier, up-cast to bf16 in their HBM slot; activations / gate / arithmetic bf16
EXPERT_SWITCH, EXPERT_SWITCH_GATED, EXPERT_NLLB, EXPERT_FSGPT, EXPERT_MIXTRAL, EXPERT_DEEPSEEK = 0, 1, 2, 3, 4, 5
ROUTER_MIXTRAL, ROUTER_DEEPSEEK, ROUTER_SWITCH, ROUTER_NLLB, ROUTER_SOFTMAX_TOPK, ROUTER_DEEPSEEK_V3 = 0, 1, 2, 3, 4, 5
POLICY_LFU_INCACHE, POLICY_LRU = 0, 1

# moe_infinity/common/constants.py:29-37 (MODEL_MAPPING_TYPES)
MODEL_MAPPING_TYPES = {"switch": EXPERT_SWITCH, "nllb": EXPERT_NLLB, "mixtral": EXPERT_MIXTRAL, "grok": EXPERT_MIXTRAL,
                       "arctic": EXPERT_MIXTRAL, "deepseek": EXPERT_DEEPSEEK, "deepseek_v3": EXPERT_DEEPSEEK}


@dataclass
class ArcherConfig:
    offload_path: str = ""
    trace_capacity: int = 1000
    trace_path: Optional[str] = None
    prefetch: bool = False
    device_memory_ratio: float = 0.9
    num_threads: int = 8  # accepted for compatibility; the HIP engine has no exec-thread pool
    host_memory_ratio: float = 0.9
    # additions
    device_memory_bytes: int = 0  # explicit expert-cache budget (needed to create misses on a 288 GB part)
    cache_policy: str = "lfu_incache"  # or "lru"

    @classmethod
    def load_from_json(cls, d):
        known = {f.name for f in fields(cls)}
        return cls(**{k: v for k, v in d.items() if k in known})

    @classmethod
    def load_from_file(cls, path):
        with open(path) as f:
            return cls.load_from_json(json.load(f))

    def __post_init__(self):
        self.perfect_cache_file = os.path.join(self.offload_path, "perfect_cache")
        if self.trace_path is not None:
            self.trace_path = os.path.abspath(self.trace_path)
            if os.path.isdir(self.trace_path):
                raise ValueError("The trace path should be a file, not a directory.")


@dataclass
class EngineConfig:
    num_layers: int
    num_experts: int
    expert_type: int
    hidden: int
    inter: int
    top_k: int
    router_kind: int
    dtype: int = DTYPE_BF16
    gate_dtype: Optional[int] = None  # defaults to dtype
    shared_inter: int = 0
    norm_topk_prob: bool = False
    routed_scaling_factor: float = 1.0
    n_group: int = 0
    topk_group: int = 0
    expert_capacity: int = 0
    device_id: int = 0
    device_memory_ratio: float = 0.9
    device_memory_bytes: int = 0
    host_memory_bytes: int = 0
    policy: int = POLICY_LFU_INCACHE
    ep_rank: int = 0
    ep_size: int = 1
    max_tokens: int = 64

    def to_dict(self):
        return asdict(self)


def mixtral_8x7b(dtype=DTYPE_BF16, **kw):
    """Mixtral-8x7B shapes: H=4096 F=14336 E=8 K=2 L=32 bf16 (dtype=DTYPE_F16: fp16 experts)."""
    return EngineConfig(num_layers=32, num_experts=8, expert_type=EXPERT_MIXTRAL, hidden=4096, inter=14336, top_k=2,
                        router_kind=ROUTER_MIXTRAL, dtype=dtype, **kw)


def deepseek_v2_lite(dtype=DTYPE_BF16, **kw):
    """DeepSeek-V2-Lite: H=2048 F=1408 E=64 K=6, 26 MoE layers, 2 shared experts (F=2816), bf16 (dtype=DTYPE_F16: fp16
    experts), greedy, norm_topk_prob=False, routed_scaling_factor=1.0."""
    return EngineConfig(num_layers=26, num_experts=64, expert_type=EXPERT_DEEPSEEK, hidden=2048, inter=1408, top_k=6,
                        router_kind=ROUTER_DEEPSEEK, dtype=dtype, shared_inter=2816, **kw)


def switch_base_8(**kw):
    """Switch-base-8: H=768 F=3072 E=8 K=1, 6+6 sparse layers, fp32, expert_capacity=64."""
    return EngineConfig(num_layers=12, num_experts=8, expert_type=EXPERT_SWITCH, hidden=768, inter=3072, top_k=1,
                        router_kind=ROUTER_SWITCH, dtype=DTYPE_F32, expert_capacity=64, **kw)


def nllb_moe_54b(dtype=DTYPE_BF16, **kw):
    """NLLB-MoE-54B: H=2048 F=8192 E=128 K=2, 6+6 sparse layers, biases."""
    return EngineConfig(num_layers=12, num_experts=128, expert_type=EXPERT_NLLB, hidden=2048, inter=8192, top_k=2,
                        router_kind=ROUTER_NLLB, dtype=dtype, gate_dtype=dtype, **kw)
