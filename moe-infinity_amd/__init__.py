"""moe-infinity_amd — MI355X-native expert-offload engine behind MoE-Infinity's MoE()/forward API.

Host-side mirror of the reference's interface for the hot path (router -> dispatch -> residency
-> grouped expert FFN -> combine) on top of the C ABI in ``include/moeinf.h`` /
``libmoeinf_hip.so`` (hand-written HIP for gfx950).  There is NO CPU fallback: every compute entry
point raises if the HIP library is missing or no GPU is visible.
"""
from ._lib import MoeInfError, lib_path, load_library  # noqa: F401
from .config import ArcherConfig, EngineConfig  # noqa: F401
from .engine import CacheSim, ExpertTracerNative, MoEEngine  # noqa: F401
from .prefetch_handle import PrefetchHandle  # noqa: F401
from . import prefetch_op  # noqa: F401  (drop-in for the reference's pybind module)

__version__ = "0.1.0"
