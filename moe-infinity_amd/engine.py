"""Python host layer over the C ABI (include/moeinf.h).

``MoEEngine`` is the object the reference's Python side talks to through
``prefetch_op.prefetch_handle`` + ``prefetch_op.expert_dispatcher``
(core/python/py_archer_prefetch.cpp:10-93) — one object here, because the HIP engine fuses
router, dispatch, residency, FFN and combine behind ``moe_forward``.  torch is used only for
device memory and streams.
"""
import ctypes as C
from typing import Dict, List, Optional, Sequence

import numpy as np
import torch

from . import _lib
from ._lib import Config, EpProfile, MoeInfError, Profile, Stats, check, load_library
from .config import DTYPE_BF16, DTYPE_F16, DTYPE_F32, DTYPE_F8E4M3, EngineConfig

FWD_DEFAULT, FWD_ROUTE_ONLY, FWD_NO_COMBINE = 0, 1, 2
_TORCH_DTYPE = {DTYPE_BF16: torch.bfloat16, DTYPE_F32: torch.float32, DTYPE_F16: torch.float16, DTYPE_F8E4M3: torch.bfloat16}  # (activations)
_ALIGN = 4096


def _ptr(t: torch.Tensor):
    return C.c_void_p(t.data_ptr())


# the raw handle of torch's current stream without building a torch.cuda.Stream object per call (the decode loop calls
# forward() once per MoE layer: tens of thousands of times per second on the small models)
_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def _current_stream_handle(device: torch.device) -> int:
    if _raw_stream is not None:
        return _raw_stream(device.index)
    return torch.cuda.current_stream(device).cuda_stream


def _i32(a):
    a = np.ascontiguousarray(a, dtype=np.int32)
    return a, a.ctypes.data_as(C.POINTER(C.c_int32))


class MoEEngine:
    """One engine per (process, GPU).  Not thread-safe (one caller thread, like the reference's
    GIL-held pybind calls)."""

    def __init__(self, cfg: EngineConfig):
        self._h = C.c_void_p()
        self.lib = load_library()
        if not torch.cuda.is_available():
            raise RuntimeError("moe-infinity_amd needs a visible MI355X (torch.cuda.is_available() is False); "
                               "there is no CPU fallback")
        self.cfg = cfg
        c = Config()
        c.abi_version = _lib.ABI_VERSION
        for name, _ in Config._fields_:
            if name in ("abi_version",):
                continue
            v = getattr(cfg, name)
            if name == "gate_dtype" and v is None:
                v = cfg.dtype
            if name == "norm_topk_prob":
                v = int(bool(v))
            setattr(c, name, v)
        check(self.lib.moeinf_create(C.byref(c), C.byref(self._h)))
        self.dtype = _TORCH_DTYPE[cfg.dtype]
        # dtype of the expert blobs in the HOST tier (pack_expert): fp8 experts travel as e4m3fn bytes and are up-cast to bf16 in their slot
        self.host_dtype = torch.float8_e4m3fn if cfg.dtype == DTYPE_F8E4M3 else self.dtype
        self.gate_dtype = _TORCH_DTYPE[cfg.dtype if cfg.gate_dtype is None else cfg.gate_dtype]
        self.device = torch.device("cuda", cfg.device_id)
        self._H, self._gate_shape = cfg.hidden, torch.Size((cfg.num_experts, cfg.hidden))
        self._moe_forward = self.lib.moeinf_moe_forward
        self._last_T = 0
        self._stores = set()  # OffloadStore objects experts were registered from (kept alive until close())

    # ---- lifecycle ---------------------------------------------------------------------------
    def close(self):
        if self._h:
            check(self.lib.moeinf_destroy(self._h))
            self._h = C.c_void_p()
            self._stores.clear()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- expert blobs ------------------------------------------------------------------------
    def expert_layout(self, which: int = 0):
        off = (C.c_int64 * 4)()
        siz = (C.c_int64 * 4)()
        n = C.c_int32()
        tot = C.c_int64()
        check(self.lib.moeinf_expert_layout(self._h, which, off, siz, C.byref(n), C.byref(tot)))
        return list(off)[: n.value], list(siz)[: n.value], tot.value

    def pack_expert(self, tensors: Sequence[torch.Tensor], which: int = 0) -> torch.Tensor:
        """Tensors in the reference's blob order -> one 4 KiB-aligned uint8 blob (CPU)."""
        off, siz, tot = self.expert_layout(which)
        if len(tensors) != len(off):
            raise ValueError(f"expected {len(off)} tensors, got {len(tensors)}")
        blob = torch.zeros(tot, dtype=torch.uint8)
        for t, o, s in zip(tensors, off, siz):
            t = t.detach().to("cpu", self.host_dtype).contiguous()
            if t.numel() * t.element_size() != s:
                raise ValueError(f"tensor of {t.numel() * t.element_size()} bytes where the layout needs {s}")
            blob[o:o + s] = t.view(torch.uint8).reshape(-1)
        return blob

    def register_expert(self, layer: int, expert: int, tensors: Optional[Sequence[torch.Tensor]] = None):
        """expert_dispatcher.register_expert + prefetch_handle.offload (host copy into the pinned arena).
        tensors=None reserves an uninitialised arena block (fill it through ``expert_host_view``)."""
        if tensors is None:
            check(self.lib.moeinf_register_expert(self._h, layer, expert, None, 0))
            return
        blob = self.pack_expert(tensors)
        check(self.lib.moeinf_register_expert(self._h, layer, expert, _ptr(blob), blob.numel()))

    def expert_host_view(self, layer: int, expert: int) -> torch.Tensor:
        """uint8 view of the expert's pinned host blob (engine-owned memory)."""
        p = C.c_void_p()
        check(self.lib.moeinf_expert_host_ptr(self._h, layer, expert, C.byref(p)))
        _, _, tot = self.expert_layout(0)
        buf = (C.c_uint8 * tot).from_address(p.value)
        return torch.frombuffer(buf, dtype=torch.uint8)

    def register_shared(self, layer: int, tensors: Sequence[torch.Tensor]):
        blob = self.pack_expert(tensors, which=1)
        check(self.lib.moeinf_register_shared(self._h, layer, _ptr(blob), blob.numel()))

    # ---- hot path ----------------------------------------------------------------------------
    def _check_dev(self, t: torch.Tensor, dtype, what):
        if not t.is_cuda or t.device.index != self.cfg.device_id:
            raise ValueError(f"{what} must live on cuda:{self.cfg.device_id}")
        if t.dtype != dtype:
            raise ValueError(f"{what} must be {dtype}, got {t.dtype}")
        if not t.is_contiguous():
            raise ValueError(f"{what} must be contiguous")

    def forward(self, layer: int, x: torch.Tensor, gate_w: torch.Tensor, batch_rows: int = 1,
                out: Optional[torch.Tensor] = None, flags: int = FWD_DEFAULT) -> Optional[torch.Tensor]:
        """One MoE layer: x [..., H] -> out [..., H] (same shape), enqueued on the current stream."""
        # One comparison per tensor on the hot path; _check_dev (which names what is wrong) runs only when it fails.  The host
        # side of a sync-free forward is 15-20 us, the same order as a whole Switch-base-8 or DeepSeek-V2-Lite layer on the GPU.
        shape = x.shape
        x2 = x if len(shape) == 2 else x.reshape(-1, shape[-1])
        dev, dt = self.device, self.dtype
        if x2.device != dev or x2.dtype is not dt or not x2.is_contiguous():
            self._check_dev(x2, dt, "x")
        if gate_w.device != dev or gate_w.dtype is not self.gate_dtype or not gate_w.is_contiguous():
            self._check_dev(gate_w, self.gate_dtype, "gate_w")
        T, H = x2.shape
        if H != self._H or gate_w.shape != self._gate_shape:
            raise ValueError("x / gate_w shape does not match the engine config")
        if out is None:
            if not (flags & FWD_ROUTE_ONLY):
                out = torch.empty_like(x2)
        elif out.device != dev or out.dtype is not dt or not out.is_contiguous():
            self._check_dev(out, dt, "out")
        rc = self._moe_forward(self._h, layer, x2.data_ptr(), T, batch_rows, gate_w.data_ptr(),
                               out.data_ptr() if out is not None else None, _current_stream_handle(dev), flags)
        if rc != 0:
            check(rc)
        self._last_T = T
        if out is None:
            return None
        return out if out.shape == shape else out.reshape(shape)

    def dispatch_mask(self, layer: int, x2: torch.Tensor, router_mask: torch.Tensor, experts: Optional[Sequence[int]] = None):
        """Grouped expert FFN for a dense router_mask[T,E] (the reference's dispatch_local contract).
        Returns (y [rows,H] expert-sorted device tensor, counts[E], hit[E]).  ``experts``: only these columns of the mask
        run (moeinf_dispatch_mask_subset: the experts enqueued on this engine), the others count as all-false."""
        self._check_dev(x2, self.dtype, "x")
        T, E = x2.shape[0], self.cfg.num_experts
        m = router_mask.reshape(T, E)
        if m.dtype == torch.bool:
            m = m.view(torch.uint8) if m.is_contiguous() else m.contiguous().view(torch.uint8)
        if m.dtype not in (torch.uint8, torch.int8, torch.int32, torch.int64):
            m = m.to(torch.uint8)
        m = m.contiguous()
        if not m.is_cuda:
            raise ValueError("router_mask must live on the engine's GPU")
        y = torch.empty((self.cfg.max_tokens * self.cfg.top_k, self.cfg.hidden), dtype=self.dtype, device=self.device)
        counts = np.empty(E, np.int32)
        hit = np.empty(E, np.int32)
        stream = C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)
        if experts is None:
            ids, n_ids = None, 0
        else:
            ids = (C.c_int32 * max(1, len(experts)))(*[int(e) for e in experts])
            n_ids = len(experts)
        check(self.lib.moeinf_dispatch_mask_subset(self._h, layer, _ptr(x2), T, _ptr(m), m.element_size(), _ptr(y),
                                                   counts.ctypes.data_as(C.POINTER(C.c_int32)),
                                                   hit.ctypes.data_as(C.POINTER(C.c_int32)), stream, ids, n_ids))
        self._last_T = T
        return y[: int(counts.sum())], counts, hit

    def combine(self, y: torch.Tensor, topk_idx: torch.Tensor, topk_w: torch.Tensor, x2: Optional[torch.Tensor] = None,
                router_prob: Optional[torch.Tensor] = None, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        """The block's combine loop on its own (moeinf_combine): y = expert-sorted expert outputs (dispatch_mask's), topk_idx
        [T,K] int32 (-1 = dropped pair), topk_w [T,K] float32, both on the device."""
        T = topk_idx.shape[0]
        self._check_dev(y, self.dtype, "y")
        self._check_dev(topk_idx, torch.int32, "topk_idx")
        self._check_dev(topk_w, torch.float32, "topk_w")
        if out is None:
            out = torch.empty((T, self.cfg.hidden), dtype=self.dtype, device=self.device)
        stream = C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)
        check(self.lib.moeinf_combine(self._h, _ptr(x2) if x2 is not None else None, _ptr(y), _ptr(topk_idx), _ptr(topk_w),
                                      _ptr(router_prob) if router_prob is not None else None, T, _ptr(out), stream))
        return out

    def routing_tensors(self, logits: bool = True, topk: bool = False):
        """Device copies of the last forward's router results (logits f32 [T,E], topk_idx i32, topk_w f32)."""
        T, K, E = self._last_T, self.cfg.top_k, self.cfg.num_experts
        lg = torch.empty((T, E), dtype=torch.float32, device=self.device) if logits else None
        ti = torch.empty((T, K), dtype=torch.int32, device=self.device) if topk else None
        tw = torch.empty((T, K), dtype=torch.float32, device=self.device) if topk else None
        stream = C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)
        check(self.lib.moeinf_copy_routing_dev(self._h, _ptr(lg) if logits else None, _ptr(ti) if topk else None,
                                               _ptr(tw) if topk else None, stream))
        return lg, ti, tw

    def routing(self) -> Dict[str, np.ndarray]:
        T, K, E = self._last_T, self.cfg.top_k, self.cfg.num_experts
        idx = np.empty(T * K, np.int32)
        w = np.empty(T * K, np.float32)
        counts = np.empty(E, np.int32)
        offsets = np.empty(E + 1, np.int32)
        slot_token = np.empty(T * K, np.int32)
        pair_slot = np.empty(T * K, np.int32)
        p = lambda a, t: a.ctypes.data_as(C.POINTER(t))  # noqa: E731
        check(self.lib.moeinf_get_routing(self._h, p(idx, C.c_int32), p(w, C.c_float), p(counts, C.c_int32),
                                          p(offsets, C.c_int32), p(slot_token, C.c_int32), p(pair_slot, C.c_int32)))
        n = int(offsets[E])
        return dict(topk_idx=idx.reshape(T, K), topk_w=w.reshape(T, K), counts=counts, offsets=offsets,
                    slot_token=slot_token[:n], pair_slot=pair_slot.reshape(T, K))

    def expert_outputs(self, rows: int) -> torch.Tensor:
        """First `rows` expert-sorted FFN output rows of the last forward (CPU tensor, engine dtype)."""
        out = torch.empty((rows, self.cfg.hidden), dtype=self.dtype)
        if rows:
            check(self.lib.moeinf_get_expert_outputs(self._h, _ptr(out), out.numel() * out.element_size()))
        return out

    def logits(self) -> np.ndarray:
        a = np.empty((self._last_T, self.cfg.num_experts), np.float32)
        check(self.lib.moeinf_get_logits(self._h, a.ctypes.data_as(C.POINTER(C.c_float)), a.size))
        return a

    # ---- cache control -----------------------------------------------------------------------
    def prefetch(self, layer: int, experts: Sequence[int], scores: Optional[Sequence[float]] = None):
        e, ep = _i32(list(experts))
        if scores is None:
            sp = None
        else:
            s = np.ascontiguousarray(scores, dtype=np.float32)
            sp = s.ctypes.data_as(C.POINTER(C.c_float))
        check(self.lib.moeinf_prefetch(self._h, layer, ep, sp, len(e)))

    def protect(self, pairs: Sequence[Sequence[int]]):
        """replace_cache_candidates: pairs = [(layer, expert), ...]"""
        l, lp = _i32([p[0] for p in pairs])
        e, ep = _i32([p[1] for p in pairs])
        check(self.lib.moeinf_protect(self._h, lp, ep, len(l)))

    def clear_expert_cache_counts(self):
        check(self.lib.moeinf_clear_cache_counts(self._h))

    def is_resident(self, layer: int, expert: int) -> bool:
        r = C.c_int32()
        check(self.lib.moeinf_is_resident(self._h, layer, expert, C.byref(r)))
        return bool(r.value)

    def sync(self):
        """wait for the last forward's stream; raises if a kernel raised the device error flag"""
        check(self.lib.moeinf_sync(self._h))

    def sync_copies(self):
        check(self.lib.moeinf_sync_copies(self._h))

    def reserve_tokens(self, max_tokens: int):
        """grow the workspace so that forwards of up to max_tokens tokens fit"""
        check(self.lib.moeinf_reserve_tokens(self._h, int(max_tokens)))
        self.cfg.max_tokens = max(self.cfg.max_tokens, int(max_tokens))

    def set_cache_budget(self, device_memory_bytes: int):
        """SetMemoryRatio at run time, in bytes (shrinking evicts by policy and frees the slots)."""
        check(self.lib.moeinf_set_cache_budget(self._h, int(device_memory_bytes)))

    def set_cache_policy(self, policy: int):
        """config.POLICY_LFU_INCACHE / POLICY_LRU from the next eviction on (moeinf_set_cache_policy)"""
        check(self.lib.moeinf_set_cache_policy(self._h, int(policy)))
        self.cfg.policy = int(policy)

    def set_prefetch_governor(self, min_useful_fraction: float = 0.5, probe_every: int = 16):
        """Stop issuing speculative copies while fewer than ``min_useful_fraction`` of the finished ones were ever
        dispatched (one probe in ``probe_every`` keeps watching); 0 switches the governor off."""
        check(self.lib.moeinf_set_prefetch_governor(self._h, float(min_useful_fraction), int(probe_every)))

    def set_predictor(self, tracer: Optional["ExpertTracerNative"], seq_id: int = -1, lookahead_layers: int = 2,
                      min_share: float = 0.15, max_experts: int = 16):
        """attach (or with tracer=None detach) the native tracer: the engine itself predicts and requests the next layers'
        experts from the routing it already holds (moeinf_set_predictor)"""
        self._predictor = tracer  # keep it alive while attached
        check(self.lib.moeinf_set_predictor(self._h, tracer._h if tracer is not None else None, int(seq_id), int(lookahead_layers),
                                            float(min_share), int(max_experts)))

    def set_gate_bias(self, layer: int, bias: Optional[torch.Tensor]):
        """DeepSeek-V3's e_score_correction_bias of one layer (moeinf_set_gate_bias): [num_experts] fp32 on the engine's device,
        borrowed (kept alive here); None = zeros"""
        if bias is not None:
            if bias.device != self.device or bias.dtype != torch.float32 or not bias.is_contiguous() or bias.numel() != self.cfg.num_experts:
                raise ValueError("set_gate_bias: a contiguous fp32 [num_experts] tensor on the engine's device")
        if not hasattr(self, "_gate_bias"):
            self._gate_bias = {}
        self._gate_bias[int(layer)] = bias
        check(self.lib.moeinf_set_gate_bias(self._h, int(layer), None if bias is None else C.c_void_p(bias.data_ptr())))

    def set_lookahead(self, gates: Optional[Sequence[torch.Tensor]], max_experts: int = 0):
        """next-layer gate lookahead (moeinf_set_lookahead): ``gates`` = every layer's gate weight [E, H] on this engine's
        device, in layer order (borrowed: keep them alive); None turns it off.  ``max_experts``: predictions issued per
        forward (default: top_k x 2)."""
        if gates is None:
            self._lookahead_gates = None
            check(self.lib.moeinf_set_lookahead(self._h, None, 0, 0))
            return
        gates = list(gates)
        if len(gates) != self.cfg.num_layers:
            raise ValueError(f"set_lookahead: {len(gates)} gates for {self.cfg.num_layers} layers")
        for g in gates:
            if g.device != self.device or not g.is_contiguous() or tuple(g.shape) != (self.cfg.num_experts, self.cfg.hidden):
                raise ValueError("set_lookahead: every gate must be a contiguous [num_experts, hidden] tensor on the engine's device")
        arr = (C.c_void_p * len(gates))(*[g.data_ptr() for g in gates])
        self._lookahead_gates = gates  # borrowed by the engine
        check(self.lib.moeinf_set_lookahead(self._h, arr, len(gates), int(max_experts) or 2 * self.cfg.top_k))

    def expert_counters(self) -> np.ndarray:
        """[L, E, 7] = visit, hit, miss, prefetch, incache_visit_count, resident, unused_count (get_hit_rate analogue)."""
        a = np.empty((self.cfg.num_layers, self.cfg.num_experts, 7), np.int64)
        check(self.lib.moeinf_get_expert_counters(self._h, a.ctypes.data_as(C.POINTER(C.c_int64)), a.size))
        return a

    def stats(self) -> dict:
        s = Stats()
        check(self.lib.moeinf_get_stats(self._h, C.byref(s)))
        return s.as_dict()

    def reset_stats(self):
        check(self.lib.moeinf_reset_stats(self._h))

    def set_profiling(self, on):
        """True / 1: per-kernel events; 2: per-phase events of ep_moe_forward; 3: both"""
        check(self.lib.moeinf_set_profiling(self._h, int(on)))
        self._profiling = int(on)

    def profile(self) -> dict:
        """Accumulated per-kernel event timings + algorithmic bytes since the last call (resets)."""
        p = Profile()
        check(self.lib.moeinf_get_profile(self._h, C.byref(p)))
        return p.as_dict()

    # ---- expert parallel ---------------------------------------------------------------------
    def ep_row_elems(self) -> int:
        n = C.c_int32()
        check(self.lib.moeinf_ep_row_elems(self._h, C.byref(n)))
        return n.value

    def ep_pack(self, x2: torch.Tensor, send: torch.Tensor, send_counts: Optional[torch.Tensor], cap_rows: int):
        stream = C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)
        check(self.lib.moeinf_ep_pack(self._h, _ptr(x2), _ptr(send), _ptr(send_counts) if send_counts is not None else None,
                                      cap_rows, stream))

    def ep_route_pack(self, layer: int, x2: torch.Tensor, gate_w: torch.Tensor, send: torch.Tensor,
                      send_counts: Optional[torch.Tensor], cap_rows: int, batch_rows: int = 1):
        """router + pack of the fixed-capacity exchange in one call (moeinf_ep_route_pack)"""
        stream = C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)
        check(self.lib.moeinf_ep_route_pack(self._h, layer, _ptr(x2), x2.shape[0], batch_rows, _ptr(gate_w), _ptr(send),
                                            _ptr(send_counts) if send_counts is not None else None, cap_rows, stream))
        self._last_T = x2.shape[0]

    def ep_expert_ffn(self, layer: int, recv: torch.Tensor, y: torch.Tensor, cap_rows: int):
        stream = C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)
        check(self.lib.moeinf_ep_expert_ffn(self._h, layer, _ptr(recv), _ptr(y), cap_rows, stream))

    def ep_pack_compact(self, x2: torch.Tensor, send: torch.Tensor, send_counts: torch.Tensor):
        stream = C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)
        check(self.lib.moeinf_ep_pack_compact(self._h, _ptr(x2), _ptr(send), _ptr(send_counts), stream))

    def ep_expert_ffn_rows(self, layer: int, recv: torch.Tensor, y: torch.Tensor, nrows: int):
        stream = C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)
        check(self.lib.moeinf_ep_expert_ffn_rows(self._h, layer, _ptr(recv), _ptr(y), nrows, stream))

    def ep_combine(self, x2: torch.Tensor, ret: torch.Tensor, out: torch.Tensor, cap_rows: int):
        stream = C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)
        check(self.lib.moeinf_ep_combine(self._h, _ptr(x2), _ptr(ret), _ptr(out), cap_rows, stream))


    # ---- native transport (RCCL called from inside the engine) ---------------------------------
    def ep_comm_available(self) -> bool:
        ok = C.c_int32()
        check(self.lib.moeinf_ep_comm_available(C.byref(ok)))
        return bool(ok.value)

    def ep_comm_unique_id(self) -> bytes:
        buf = (C.c_uint8 * 128)()
        check(self.lib.moeinf_ep_comm_unique_id(buf, 128))
        return bytes(buf)

    def ep_comm_prepare(self, cap_tokens: int):
        """the local half of the RCCL bootstrap (buffers, library binding): no collective inside"""
        check(self.lib.moeinf_ep_comm_prepare(self._h, int(cap_tokens)))

    def ep_comm_init(self, unique_id: bytes, cap_tokens: int):
        buf = (C.c_uint8 * 128).from_buffer_copy(unique_id)
        check(self.lib.moeinf_ep_comm_init(self._h, buf, 128, int(cap_tokens)))

    def ep_all_to_all(self, send: torch.Tensor, recv: torch.Tensor):
        """equal-split all-to-all of two device tensors of the same size (segment p goes to rank p)"""
        nbytes = send.numel() * send.element_size()
        if nbytes != recv.numel() * recv.element_size() or nbytes % self.cfg.ep_size:
            raise ValueError("send/recv must have the same size, divisible by ep_size")
        stream = C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)
        check(self.lib.moeinf_ep_all_to_all(self._h, _ptr(send), _ptr(recv), nbytes // self.cfg.ep_size, stream))

    def ep_moe_forward(self, layer: int, x2: torch.Tensor, gate_w: torch.Tensor, out: torch.Tensor, batch_rows: int = 1):
        """one expert-parallel MoE layer in one host call (moeinf_ep_moe_forward)"""
        stream = C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)
        check(self.lib.moeinf_ep_moe_forward(self._h, layer, _ptr(x2), x2.shape[0], batch_rows, _ptr(gate_w), _ptr(out), stream))
        self._last_T = x2.shape[0]

    # ---- direct peer-store exchange (no collective; include/moeinf.h: moeinf_ep_peer_*) ---------
    PEER_BLOB_BYTES = 192
    EP_TRANSPORTS = {0: "none", 1: "rccl", 2: "peer-store"}

    def ep_peer_export(self, cap_tokens: int) -> bytes:
        """allocate this rank's exchange window; returns the 192-byte blob the other ranks need to map it"""
        buf = (C.c_uint8 * self.PEER_BLOB_BYTES)()
        check(self.lib.moeinf_ep_peer_export(self._h, int(cap_tokens), buf, self.PEER_BLOB_BYTES))
        return bytes(buf)

    def ep_peer_attach(self, blobs: bytes):
        """map every rank's window (blobs of all ranks, concatenated in rank order)"""
        buf = (C.c_uint8 * len(blobs)).from_buffer_copy(blobs)
        check(self.lib.moeinf_ep_peer_attach(self._h, buf, len(blobs)))

    def ep_peer_get_timeout_ms(self) -> int:
        ms = C.c_int(0)
        check(self.lib.moeinf_ep_peer_get_timeout_ms(self._h, C.byref(ms)))
        return int(ms.value)

    def ep_peer_set_timeout_ms(self, ms: int) -> int:
        """returns the timeout that was in force (what a caller that shortens it temporarily restores)"""
        prev = self.ep_peer_get_timeout_ms()
        check(self.lib.moeinf_ep_peer_set_timeout_ms(self._h, int(ms)))
        return prev

    def ep_peer_release(self):
        """moeinf_ep_peer_release: unmap the peers and free the window (the group agreed not to use this transport)"""
        check(self.lib.moeinf_ep_peer_release(self._h))

    def ep_peer_selftest(self) -> bool:
        """tagged rows + flags to and from every peer (every rank must call it); False on a mismatch or a timeout"""
        ok = C.c_int32()
        stream = C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)
        check(self.lib.moeinf_ep_peer_selftest(self._h, stream, C.byref(ok)))
        return bool(ok.value)

    def ep_select_transport(self, name: str):
        check(self.lib.moeinf_ep_select_transport(self._h, {"rccl": 1, "peer-store": 2}[name]))

    def ep_set_uniform_tokens(self, on: bool = True):
        """every rank passes the same token count to every ep_moe_forward: one-token forwards take the broadcast form"""
        check(self.lib.moeinf_ep_set_uniform_tokens(self._h, 1 if on else 0))

    def ep_transport(self) -> dict:
        o = (C.c_int32 * 4)()
        check(self.lib.moeinf_ep_transport(self._h, o))
        return {"transport": self.EP_TRANSPORTS.get(o[0], "?"), "shared_device": bool(o[1]), "poll_in_kernels": bool(o[2]), "exchanges": int(o[3])}

    def ep_profile(self) -> dict:
        p = EpProfile()
        check(self.lib.moeinf_ep_get_profile(self._h, C.byref(p)))
        return p.as_dict()


class CacheSim:
    """Host-only replacement-policy simulator (same code the engine uses for eviction)."""

    def __init__(self, num_slots: int, policy: int = 0):
        self.lib = load_library()
        self._h = C.c_void_p()
        check(self.lib.moeinf_cache_sim_create(num_slots, policy, C.byref(self._h)))

    def access(self, ident: int):
        hit = C.c_int32()
        ev = C.c_int64()
        check(self.lib.moeinf_cache_sim_access(self._h, ident, C.byref(hit), C.byref(ev)))
        return bool(hit.value), ev.value

    def protect(self, ids: Sequence[int]):
        a = np.ascontiguousarray(list(ids), dtype=np.int64)
        check(self.lib.moeinf_cache_sim_protect(self._h, a.ctypes.data_as(C.POINTER(C.c_int64)), len(a)))

    def clear_counts(self):
        check(self.lib.moeinf_cache_sim_clear_counts(self._h))

    def __del__(self):
        try:
            if self._h:
                self.lib.moeinf_cache_sim_destroy(self._h)
        except Exception:
            pass


class ExpertTracerNative:
    """Native tracer/predictor (moe_infinity/memory/expert_tracer.py + expert_predictor.py +
    expert_prefetcher.py ordering).  Host only."""

    def __init__(self, num_layers: int, num_experts: int, capacity: int):
        self.lib = load_library()
        self.L, self.E, self.capacity = num_layers, num_experts, capacity
        self._h = C.c_void_p()
        check(self.lib.moeinf_tracer_create(num_layers, num_experts, capacity, C.byref(self._h)))

    def load_trace(self, eams: np.ndarray):
        a = np.ascontiguousarray(eams, dtype=np.float32)
        assert a.shape[1:] == (self.L, self.E)
        check(self.lib.moeinf_tracer_load(self._h, a.ctypes.data_as(C.POINTER(C.c_float)), a.shape[0]))

    def create_entry(self) -> int:
        s = C.c_int64()
        check(self.lib.moeinf_tracer_create_entry(self._h, C.byref(s)))
        return s.value

    def finish_entry(self, seq_id: int):
        check(self.lib.moeinf_tracer_finish_entry(self._h, seq_id))

    def predict(self, seq_id: int, expert_list, layer_idx: int):
        e, ep = _i32(np.asarray(expert_list).reshape(-1))
        m = np.empty((self.L, self.E), np.float32)
        near = C.c_int32()
        check(self.lib.moeinf_tracer_predict(self._h, seq_id, layer_idx, ep, len(e), m.ctypes.data_as(C.POINTER(C.c_float)),
                                             C.byref(near)))
        return m, near.value

    def prefetch_order(self, layer_idx: int, matrix: np.ndarray):
        m = np.ascontiguousarray(matrix, dtype=np.float32)
        n = C.c_int32()
        ls = np.empty(self.L * self.E, np.int32)
        es = np.empty(self.L * self.E, np.int32)
        sc = np.empty(self.L * self.E, np.float32)
        check(self.lib.moeinf_tracer_prefetch_order(self._h, layer_idx, m.ctypes.data_as(C.POINTER(C.c_float)),
                                                    ls.ctypes.data_as(C.POINTER(C.c_int32)),
                                                    es.ctypes.data_as(C.POINTER(C.c_int32)),
                                                    sc.ctypes.data_as(C.POINTER(C.c_float)), C.byref(n)))
        return ls[: n.value], es[: n.value], sc[: n.value]

    def get_eam(self, seq_id: int) -> np.ndarray:
        a = np.empty((self.L, self.E), np.float64)
        check(self.lib.moeinf_tracer_get_eam(self._h, seq_id, a.ctypes.data_as(C.POINTER(C.c_double))))
        return a

    def __del__(self):
        try:
            if self._h:
                self.lib.moeinf_tracer_destroy(self._h)
        except Exception:
            pass
