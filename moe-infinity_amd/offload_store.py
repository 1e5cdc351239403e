"""Disk tier: the reference's offload directory (``archer_index`` + ``archer_param_<n>``), read and
written in its own format (include/moeinf.h "disk tier"; core/aio/* in the reference).  Host only."""
import ctypes as C
from typing import Dict, List, Sequence

import numpy as np
import torch

from ._lib import check, load_library

# c10::ScalarType codes (the dtype byte of the index's options block)
SCALAR_TYPE = {torch.uint8: 0, torch.int8: 1, torch.int16: 2, torch.int32: 3, torch.int64: 4, torch.float16: 5,
               torch.float32: 6, torch.float64: 7, torch.bool: 11, torch.bfloat16: 15,
               torch.float8_e5m2: 23, torch.float8_e4m3fn: 24}  # (c10/core/ScalarType.h: Float8_e5m2 = 23, Float8_e4m3fn = 24)
TORCH_DTYPE = {v: k for k, v in SCALAR_TYPE.items()}


class OffloadStore:
    def __init__(self, offload_path: str):
        self.lib = load_library()
        self._h = C.c_void_p()
        check(self.lib.moeinf_store_open(str(offload_path).encode(), C.byref(self._h)))

    def close(self):
        if self._h:
            check(self.lib.moeinf_store_close(self._h))
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __len__(self):
        n = C.c_int64()
        check(self.lib.moeinf_store_count(self._h, C.byref(n)))
        return n.value

    def ids(self) -> List[int]:
        n = len(self)
        a = (C.c_uint32 * max(n, 1))()
        check(self.lib.moeinf_store_ids(self._h, a, n))
        return list(a)[:n]

    def offload(self, tensor: torch.Tensor, tensor_id: int):
        """prefetch_handle.offload(tensor, id)"""
        t = tensor.detach().cpu().contiguous()
        dims = (C.c_int64 * max(t.dim(), 1))(*t.shape)
        check(self.lib.moeinf_store_put(self._h, tensor_id, C.c_void_p(t.data_ptr()), t.numel() * t.element_size(), dims,
                                        t.dim(), SCALAR_TYPE[t.dtype]))

    def flush(self):
        check(self.lib.moeinf_store_flush(self._h))

    def meta(self, tensor_id: int):
        found, nd, st = C.c_int32(), C.c_int32(), C.c_int32()
        nbytes, off = C.c_uint64(), C.c_int64()
        dims = (C.c_int64 * 8)()
        check(self.lib.moeinf_store_meta(self._h, tensor_id, C.byref(found), C.byref(nbytes), C.byref(off), C.byref(nd),
                                         dims, C.byref(st)))
        if not found.value:
            return None
        return dict(nbytes=nbytes.value, offset=off.value, shape=tuple(dims[: nd.value]), scalar_type=st.value)

    def is_tensor_offloaded(self, tensor_id: int) -> bool:
        return self.meta(tensor_id) is not None

    def load(self, tensor_id: int) -> torch.Tensor:
        m = self.meta(tensor_id)
        if m is None:
            raise KeyError(tensor_id)
        dt = TORCH_DTYPE[m["scalar_type"]]
        t = torch.empty(m["shape"], dtype=dt)
        check(self.lib.moeinf_store_get(self._h, tensor_id, C.c_void_p(t.data_ptr()), m["nbytes"]))
        return t

    def load_to_device(self, tensor_id: int, dst: torch.Tensor):
        """disk -> device for a dense tensor: dst is a contiguous CUDA tensor (any dtype) with room for the payload;
        the copy is ordered on the current stream."""
        if not dst.is_cuda or not dst.is_contiguous():
            raise ValueError("dst must be a contiguous CUDA tensor")
        stream = C.c_void_p(torch.cuda.current_stream(dst.device).cuda_stream)
        check(self.lib.moeinf_store_get_device(self._h, tensor_id, C.c_void_p(dst.data_ptr()), dst.numel() * dst.element_size(), stream))

    def register_expert(self, engine, layer: int, expert: int, tensor_ids: Sequence[int]):
        """disk -> pinned host arena for one expert (tensor ids in the reference's blob order)."""
        a = (C.c_uint32 * len(tensor_ids))(*tensor_ids)
        check(self.lib.moeinf_register_expert_from_store(engine._h, layer, expert, self._h, a, len(tensor_ids)))
        # the engine may re-read this expert from the directory for as long as it lives: it keeps the store alive (the
        # C side refuses moeinf_store_close while experts of a live engine are backed by it)
        engine._stores.add(self)
