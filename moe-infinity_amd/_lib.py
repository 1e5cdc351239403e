"""ctypes binding of libmoeinf_hip.so (include/moeinf.h).  Loading never falls back to
anything else: a missing library is an ImportError-class failure at first use."""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None
ABI_VERSION = 4  # include/moeinf.h MOEINF_ABI_VERSION


class MoeInfError(RuntimeError):
    """Raised for every non-zero status of the C ABI; message = moeinf_last_error()."""

    def __init__(self, code, msg):
        super().__init__(f"moeinf error {code}: {msg}")
        self.code = code


def lib_path():
    return os.path.join(HERE, "libmoeinf_hip.so")


class Config(C.Structure):
    _fields_ = [
        ("abi_version", C.c_int32), ("device_id", C.c_int32), ("num_layers", C.c_int32), ("num_experts", C.c_int32),
        ("expert_type", C.c_int32), ("dtype", C.c_int32), ("hidden", C.c_int32), ("inter", C.c_int32),
        ("shared_inter", C.c_int32), ("top_k", C.c_int32), ("router_kind", C.c_int32), ("gate_dtype", C.c_int32),
        ("norm_topk_prob", C.c_int32), ("routed_scaling_factor", C.c_float), ("n_group", C.c_int32),
        ("topk_group", C.c_int32), ("expert_capacity", C.c_int32), ("device_memory_ratio", C.c_double),
        ("device_memory_bytes", C.c_int64), ("host_memory_bytes", C.c_int64), ("policy", C.c_int32),
        ("ep_rank", C.c_int32), ("ep_size", C.c_int32), ("max_tokens", C.c_int32),
    ]


class Stats(C.Structure):
    _fields_ = [
        ("forwards", C.c_int64), ("expert_hits", C.c_int64), ("expert_misses", C.c_int64),
        ("prefetch_issued", C.c_int64), ("prefetch_useful", C.c_int64), ("evictions", C.c_int64),
        ("h2d_bytes", C.c_int64), ("slots_total", C.c_int64), ("slots_used", C.c_int64), ("slot_bytes", C.c_int64),
        ("host_arena_bytes", C.c_int64), ("h2d_busy_ms", C.c_double), ("exposed_wait_ms", C.c_double),
        ("prefetch_queued", C.c_int64), ("prefetch_cancelled", C.c_int64), ("prefetch_dropped", C.c_int64),
        ("prefetch_wasted", C.c_int64), ("inflight_hits", C.c_int64), ("host_evictions", C.c_int64),
        ("disk_reads", C.c_int64), ("disk_bytes", C.c_int64), ("disk_reads_async", C.c_int64),
        ("prefetch_throttled", C.c_int64),
    ]

    def as_dict(self):
        return {n: getattr(self, n) for n, _ in self._fields_}


class Profile(C.Structure):
    _fields_ = [
        ("forwards", C.c_int64), ("ffn1_launches", C.c_int64), ("ffn2_launches", C.c_int64),
        ("ffn1_bytes", C.c_int64), ("ffn2_bytes", C.c_int64), ("route_bytes", C.c_int64), ("combine_bytes", C.c_int64),
        ("route_ms", C.c_double), ("ffn1_ms", C.c_double), ("ffn2_ms", C.c_double), ("combine_ms", C.c_double),
        ("host_wait_ms", C.c_double), ("fused_layers", C.c_int64), ("kernel_timed_launches", C.c_int64),
    ]

    def as_dict(self):
        return {n: getattr(self, n) for n, _ in self._fields_}


class EpProfile(C.Structure):
    _fields_ = [("calls", C.c_int64), ("route_pack_ms", C.c_double), ("a2a_dispatch_ms", C.c_double), ("owner_ffn_ms", C.c_double),
                ("a2a_combine_ms", C.c_double), ("combine_ms", C.c_double)]

    def as_dict(self):
        return {n: getattr(self, n) for n, _ in self._fields_}


_P = C.c_void_p
_I32P = C.POINTER(C.c_int32)
_I64P = C.POINTER(C.c_int64)
_F32P = C.POINTER(C.c_float)
_F64P = C.POINTER(C.c_double)

# name -> (restype, argtypes); every symbol include/moeinf.h declares
PROTOTYPES = {
    "moeinf_last_error": (C.c_char_p, []),
    "moeinf_abi_version": (C.c_int, []),
    "moeinf_ffn_ring2_form": (C.c_int, [C.c_int] * 8 + [_I32P]),
    "moeinf_set_cache_policy": (C.c_int, [_P, C.c_int]),
    "moeinf_rows_estimate": (C.c_int, [C.c_int] * 3),
    "moeinf_fence_ring": (C.c_int, []),
    "moeinf_fence_cover_pos": (C.c_int, [C.POINTER(C.c_uint64), C.c_uint64, C.c_uint64]),
    "moeinf_create": (C.c_int, [C.POINTER(Config), C.POINTER(_P)]),
    "moeinf_destroy": (C.c_int, [_P]),
    "moeinf_expert_layout": (C.c_int, [_P, C.c_int, _I64P, _I64P, _I32P, _I64P]),
    "moeinf_register_expert": (C.c_int, [_P, C.c_int, C.c_int, _P, C.c_int64]),
    "moeinf_expert_host_ptr": (C.c_int, [_P, C.c_int, C.c_int, C.POINTER(_P)]),
    "moeinf_register_shared": (C.c_int, [_P, C.c_int, _P, C.c_int64]),
    "moeinf_moe_forward": (C.c_int, [_P, C.c_int, _P, C.c_int, C.c_int, _P, _P, _P, C.c_uint32]),
    "moeinf_dispatch_mask": (C.c_int, [_P, C.c_int, _P, C.c_int, _P, C.c_int, _P, _I32P, _I32P, _P]),
    "moeinf_dispatch_mask_subset": (C.c_int, [_P, C.c_int, _P, C.c_int, _P, C.c_int, _P, _I32P, _I32P, _P, _I32P, C.c_int]),
    "moeinf_copy_routing_dev": (C.c_int, [_P, _P, _P, _P, _P]),
    "moeinf_get_routing": (C.c_int, [_P, _I32P, _F32P, _I32P, _I32P, _I32P, _I32P]),
    "moeinf_get_expert_outputs": (C.c_int, [_P, _P, C.c_int64]),
    "moeinf_get_logits": (C.c_int, [_P, _F32P, C.c_int64]),
    "moeinf_prefetch": (C.c_int, [_P, C.c_int, _I32P, _F32P, C.c_int]),
    "moeinf_protect": (C.c_int, [_P, _I32P, _I32P, C.c_int]),
    "moeinf_clear_cache_counts": (C.c_int, [_P]),
    "moeinf_is_resident": (C.c_int, [_P, C.c_int, C.c_int, _I32P]),
    "moeinf_sync_copies": (C.c_int, [_P]),
    "moeinf_set_cache_budget": (C.c_int, [_P, C.c_int64]),
    "moeinf_set_prefetch_governor": (C.c_int, [_P, C.c_float, C.c_int]),
    "moeinf_reserve_tokens": (C.c_int, [_P, C.c_int]),
    "moeinf_get_expert_counters": (C.c_int, [_P, _I64P, C.c_int64]),
    "moeinf_get_stats": (C.c_int, [_P, C.POINTER(Stats)]),
    "moeinf_reset_stats": (C.c_int, [_P]),
    "moeinf_set_profiling": (C.c_int, [_P, C.c_int]),
    "moeinf_get_profile": (C.c_int, [_P, C.POINTER(Profile)]),
    "moeinf_tracer_create": (C.c_int, [C.c_int, C.c_int, C.c_int, C.POINTER(_P)]),
    "moeinf_tracer_destroy": (C.c_int, [_P]),
    "moeinf_tracer_load": (C.c_int, [_P, _F32P, C.c_int]),
    "moeinf_tracer_create_entry": (C.c_int, [_P, _I64P]),
    "moeinf_tracer_finish_entry": (C.c_int, [_P, C.c_int64]),
    "moeinf_tracer_predict": (C.c_int, [_P, C.c_int64, C.c_int, _I32P, C.c_int, _F32P, _I32P]),
    "moeinf_tracer_prefetch_order": (C.c_int, [_P, C.c_int, _F32P, _I32P, _I32P, _F32P, _I32P]),
    "moeinf_tracer_get_eam": (C.c_int, [_P, C.c_int64, _F64P]),
    "moeinf_set_predictor": (C.c_int, [_P, _P, C.c_int64, C.c_int, C.c_float, C.c_int]),
    "moeinf_set_lookahead": (C.c_int, [_P, C.POINTER(C.c_void_p), C.c_int, C.c_int]),
    "moeinf_set_gate_bias": (C.c_int, [_P, C.c_int, _P]),
    "moeinf_store_open": (C.c_int, [C.c_char_p, C.POINTER(_P)]),
    "moeinf_store_close": (C.c_int, [_P]),
    "moeinf_store_put": (C.c_int, [_P, C.c_uint32, _P, C.c_uint64, _I64P, C.c_int, C.c_int]),
    "moeinf_store_flush": (C.c_int, [_P]),
    "moeinf_store_count": (C.c_int, [_P, _I64P]),
    "moeinf_store_ids": (C.c_int, [_P, C.POINTER(C.c_uint32), C.c_int64]),
    "moeinf_store_meta": (C.c_int, [_P, C.c_uint32, _I32P, C.POINTER(C.c_uint64), _I64P, _I32P, _I64P, _I32P]),
    "moeinf_store_get": (C.c_int, [_P, C.c_uint32, _P, C.c_uint64]),
    "moeinf_store_get_device": (C.c_int, [_P, C.c_uint32, _P, C.c_uint64, _P]),
    "moeinf_register_expert_from_store": (C.c_int, [_P, C.c_int, C.c_int, _P, C.POINTER(C.c_uint32), C.c_int]),
    "moeinf_cache_sim_create": (C.c_int, [C.c_int, C.c_int, C.POINTER(_P)]),
    "moeinf_cache_sim_destroy": (C.c_int, [_P]),
    "moeinf_cache_sim_access": (C.c_int, [_P, C.c_int64, _I32P, _I64P]),
    "moeinf_cache_sim_protect": (C.c_int, [_P, _I64P, C.c_int]),
    "moeinf_cache_sim_clear_counts": (C.c_int, [_P]),
    "moeinf_aio_create": (C.c_int, [C.c_int, C.c_int64, C.POINTER(_P)]),
    "moeinf_aio_destroy": (C.c_int, [_P]),
    "moeinf_aio_submit_read": (C.c_int, [_P, C.c_char_p, _P, C.c_int64, C.c_int64, C.c_int, C.c_int, _I64P]),
    "moeinf_aio_promote": (C.c_int, [_P, C.c_int64]),
    "moeinf_aio_done": (C.c_int, [_P, C.c_int64, _I32P]),
    "moeinf_aio_wait": (C.c_int, [_P, C.c_int64]),
    "moeinf_aio_stats": (C.c_int, [_P, _I64P]),
    "moeinf_pq_create": (C.c_int, [C.POINTER(_P)]),
    "moeinf_pq_destroy": (C.c_int, [_P]),
    "moeinf_pq_enqueue": (C.c_int, [_P, C.c_int64, C.c_int, C.c_int, C.c_int, _I32P]),
    "moeinf_pq_on_demand": (C.c_int, [_P, C.c_int64, C.c_int, _I32P]),
    "moeinf_pq_fetch": (C.c_int, [_P, C.c_int64, C.c_int, C.c_int, _I32P]),
    "moeinf_pq_clear_prefetch": (C.c_int, [_P, _I32P]),
    "moeinf_pq_pop": (C.c_int, [_P, _I64P, _I32P, _I32P, _I32P]),
    "moeinf_pq_snapshot": (C.c_int, [_P, _I64P, _I32P, _I32P, C.c_int, _I32P]),
    "moeinf_priority_from_score": (C.c_int, [C.c_float, _I32P]),
    "moeinf_sync": (C.c_int, [_P]),
    "moeinf_ep_row_elems": (C.c_int, [_P, _I32P]),
    "moeinf_ep_pack": (C.c_int, [_P, _P, _P, _P, C.c_int, _P]),
    "moeinf_ep_expert_ffn": (C.c_int, [_P, C.c_int, _P, _P, C.c_int, _P]),
    "moeinf_ep_pack_compact": (C.c_int, [_P, _P, _P, _P, _P]),
    "moeinf_ep_expert_ffn_rows": (C.c_int, [_P, C.c_int, _P, _P, C.c_int, _P]),
    "moeinf_ep_combine": (C.c_int, [_P, _P, _P, _P, C.c_int, _P]),
    "moeinf_ep_route_pack": (C.c_int, [_P, C.c_int, _P, C.c_int, C.c_int, _P, _P, _P, C.c_int, _P]),
    "moeinf_combine": (C.c_int, [_P, _P, _P, _P, _P, _P, C.c_int, _P, _P]),
    "moeinf_ep_comm_available": (C.c_int, [_I32P]),
    "moeinf_ep_comm_unique_id": (C.c_int, [_P, C.c_int]),
    "moeinf_ep_comm_prepare": (C.c_int, [_P, C.c_int]),
    "moeinf_ep_comm_init": (C.c_int, [_P, _P, C.c_int, C.c_int]),
    "moeinf_ep_peer_export": (C.c_int, [_P, C.c_int, _P, C.c_int]),
    "moeinf_ep_peer_attach": (C.c_int, [_P, _P, C.c_int]),
    "moeinf_ep_peer_selftest": (C.c_int, [_P, _P, _I32P]),
    "moeinf_ep_peer_release": (C.c_int, [_P]),
    "moeinf_ep_peer_set_timeout_ms": (C.c_int, [_P, C.c_int]),
    "moeinf_ep_peer_get_timeout_ms": (C.c_int, [_P, C.POINTER(C.c_int)]),
    "moeinf_ep_transport": (C.c_int, [_P, _I32P]),
    "moeinf_ep_select_transport": (C.c_int, [_P, C.c_int]),
    "moeinf_ep_set_uniform_tokens": (C.c_int, [_P, C.c_int]),
    "moeinf_ep_all_to_all": (C.c_int, [_P, _P, _P, C.c_int64, _P]),
    "moeinf_ep_moe_forward": (C.c_int, [_P, C.c_int, _P, C.c_int, C.c_int, _P, _P, _P]),
    "moeinf_ep_get_profile": (C.c_int, [_P, C.POINTER(EpProfile)]),
}


def load_library():
    """dlopen libmoeinf_hip.so (built by ``python moe-infinity_amd/build.py`` /
    ``__graft_entry__.build()``).  Raises if it is missing — there is no other backend."""
    global _LIB
    if _LIB is not None:
        return _LIB
    path = lib_path()
    if not os.path.exists(path):
        raise ImportError(f"{path} not found: build it with `python moe-infinity_amd/build.py` "
                          "(hipcc --offload-arch=gfx950). moe-infinity_amd has no CPU/PyTorch fallback.")
    lib = C.CDLL(path)
    for name, (res, args) in PROTOTYPES.items():
        fn = getattr(lib, name)  # AttributeError if the .so does not export a declared symbol
        fn.restype = res
        fn.argtypes = args
    if lib.moeinf_abi_version() != ABI_VERSION:
        raise ImportError("libmoeinf_hip.so ABI version mismatch")
    _LIB = lib
    return lib


def check(rc):
    if rc != 0:
        msg = load_library().moeinf_last_error()
        raise MoeInfError(rc, msg.decode() if msg else "")
