"""ctypes access to oracle/_ref/libmoeinf_ref.so — the REFERENCE'S OWN expert FFN modules and archer_index serializer,
built by oracle/build_ref.py from the sources under /root/reference.  TEST INFRASTRUCTURE ONLY."""
import ctypes as C
import os

import numpy as np
import torch

from . import build_ref

_LIB = None
_DT = {torch.bfloat16: 0, torch.float32: 1, torch.float16: 2}  # core/parallel/expert_module.h:20-23


def available() -> bool:
    return build_ref.available()


def lib():
    global _LIB
    if _LIB is None:
        _LIB = C.CDLL(build_ref.LIB)
    return _LIB


def expert_ffn(x: torch.Tensor, tensors, expert_type: int) -> torch.Tensor:
    """<reference module>.forward(x) for one expert (core/parallel/expert_module.cpp), tensors in blob order."""
    x = x.contiguous()
    ts = [t.contiguous() for t in tensors]
    T, H = x.shape
    F = ts[0].shape[0]
    arr = (C.c_void_p * len(ts))(*[t.data_ptr() for t in ts])
    y = torch.empty(T, H, dtype=x.dtype)
    rc = lib().ref_expert_ffn(expert_type, _DT[x.dtype], C.c_void_p(x.data_ptr()), C.c_int64(T), C.c_int64(H), C.c_int64(F),
                              arr, len(ts), C.c_void_p(y.data_ptr()))
    if rc != 0:
        raise RuntimeError(f"ref_expert_ffn failed with {rc}")
    return y


def index_write(path: str, entries: dict):
    """ArcherTensorIndex::Serialize.  entries: {id: dict(file_id, offset, size, shape, dtype)} (dtype = c10::ScalarType code)"""
    n = len(entries)
    ids = np.array(list(entries), np.uint32)
    fid = np.array([entries[i]["file_id"] for i in entries], np.uint32)
    off = np.array([entries[i]["offset"] for i in entries], np.int64)
    siz = np.array([entries[i]["size"] for i in entries], np.uint64)
    nd = np.array([len(entries[i]["shape"]) for i in entries], np.int32)
    dims = np.zeros((n, 8), np.int64)
    for r, i in enumerate(entries):
        dims[r, :len(entries[i]["shape"])] = entries[i]["shape"]
    st = np.array([entries[i]["dtype"] for i in entries], np.int32)
    p = lambda a, t: a.ctypes.data_as(C.POINTER(t))  # noqa: E731
    rc = lib().ref_index_write(path.encode(), n, p(ids, C.c_uint32), p(fid, C.c_uint32), p(off, C.c_int64), p(siz, C.c_uint64),
                               p(nd, C.c_int32), p(dims, C.c_int64), p(st, C.c_int32))
    if rc != 0:
        raise RuntimeError(f"ref_index_write failed with {rc}")


def index_read(path: str, capacity: int = 4096) -> dict:
    """ArcherTensorIndex::Deserialize -> {id: dict(file_id, offset, size, shape, dtype)}"""
    n = C.c_int32()
    ids, fid = np.zeros(capacity, np.uint32), np.zeros(capacity, np.uint32)
    off, siz = np.zeros(capacity, np.int64), np.zeros(capacity, np.uint64)
    nd, st = np.zeros(capacity, np.int32), np.zeros(capacity, np.int32)
    dims = np.zeros((capacity, 8), np.int64)
    p = lambda a, t: a.ctypes.data_as(C.POINTER(t))  # noqa: E731
    rc = lib().ref_index_read(path.encode(), capacity, C.byref(n), p(ids, C.c_uint32), p(fid, C.c_uint32), p(off, C.c_int64),
                              p(siz, C.c_uint64), p(nd, C.c_int32), p(dims, C.c_int64), p(st, C.c_int32))
    if rc != 0:
        raise RuntimeError(f"ref_index_read failed with {rc}")
    return {int(ids[i]): dict(file_id=int(fid[i]), offset=int(off[i]), size=int(siz[i]), shape=[int(v) for v in dims[i, :nd[i]]],
                              dtype=int(st[i])) for i in range(min(n.value, capacity))}
