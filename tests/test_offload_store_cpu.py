"""Disk tier: libmoeinf_hip.so reads/writes the reference's offload directory
format.  Checked against an independent struct-level restatement (oracle/offload_format_ref.py) in both
directions.  CPU only."""
import os

import numpy as np
import pytest
import torch

from oracle.offload_format_ref import read_index, write_index


def _tensors():
    g = torch.Generator().manual_seed(3)
    return {
        7: torch.randn(33, 17, generator=g).to(torch.bfloat16),   # 1122 B: not a multiple of 4096
        1: torch.randn(64, 64, generator=g),                        # 16 KiB exactly aligned
        40: torch.randn(5000, generator=g).to(torch.float16),
        3: torch.arange(12, dtype=torch.int64).reshape(3, 4),
        9: torch.randn(40, 24, generator=g).to(torch.float8_e4m3fn),  # the reference's expert dtype id 3 (ScalarType 24)
    }


def test_written_store_is_the_reference_format(tmp_path):
    from moe_infinity_amd.offload_store import SCALAR_TYPE, OffloadStore

    st = OffloadStore(str(tmp_path))
    ts = _tensors()
    for k, t in ts.items():
        st.offload(t, k)
    st.flush()
    idx = read_index(os.path.join(tmp_path, "archer_index"))
    assert set(idx) == set(ts)
    raw = open(os.path.join(tmp_path, "archer_param_0"), "rb").read()
    expect_off = 0
    for k, t in ts.items():  # offsets advance in store order, 4 KiB aligned (archer_tensor_handle.cpp:64-77)
        m = idx[k]
        assert m["file_id"] == 0 and m["offset"] == expect_off and m["offset"] % 4096 == 0
        assert m["size"] == t.numel() * t.element_size() and m["shape"] == list(t.shape)
        assert m["dtype"] == SCALAR_TYPE[t.dtype] and m["device_type"] == 0 and m["layout"] == 0
        assert raw[m["offset"]:m["offset"] + m["size"]] == t.contiguous().view(torch.uint8).numpy().tobytes()
        expect_off += (m["size"] + 4095) // 4096 * 4096
    st.close()


def test_reads_a_store_written_by_the_format_oracle(tmp_path):
    from moe_infinity_amd.offload_store import SCALAR_TYPE, OffloadStore

    ts = _tensors()
    entries, blob, off = {}, bytearray(), 0
    for k, t in ts.items():
        b = t.contiguous().view(torch.uint8).numpy().tobytes()
        entries[k] = dict(file_id=0, offset=off, size=len(b), shape=list(t.shape), dtype=SCALAR_TYPE[t.dtype],
                          pinned=False, requires_grad=False, device_index=-1, device_type=0, layout=0)
        blob += b + bytes((-len(b)) % 4096)
        off += (len(b) + 4095) // 4096 * 4096
    write_index(os.path.join(tmp_path, "archer_index"), entries)
    open(os.path.join(tmp_path, "archer_param_0"), "wb").write(bytes(blob))
    st = OffloadStore(str(tmp_path))
    assert len(st) == len(ts) and sorted(st.ids()) == sorted(ts)
    assert st.is_tensor_offloaded(7) and not st.is_tensor_offloaded(8)
    for k, t in ts.items():
        got = st.load(k)
        assert got.dtype == t.dtype and got.shape == t.shape and torch.equal(got, t)
    # appending after reopen continues at the end of the param file
    extra = torch.ones(10)
    st.offload(extra, 99)
    st.flush()
    assert read_index(os.path.join(tmp_path, "archer_index"))[99]["offset"] == off
    assert torch.equal(st.load(99), extra)
    st.close()


def test_reopen_and_rewrite_in_place(tmp_path):
    from moe_infinity_amd import MoeInfError
    from moe_infinity_amd.offload_store import OffloadStore

    st = OffloadStore(str(tmp_path))
    st.offload(torch.zeros(100), 5)
    st.close()  # close flushes
    st = OffloadStore(str(tmp_path))
    assert st.is_tensor_offloaded(5)
    st.offload(torch.full((100,), 2.0), 5)  # same id, same size: rewritten in place (archer_tensor_handle.cpp:67-74)
    assert torch.equal(st.load(5), torch.full((100,), 2.0))
    with pytest.raises(MoeInfError):
        st.offload(torch.zeros(101), 5)  # size mismatch is an error, not an abort
    with pytest.raises(KeyError):
        st.load(1234)
    st.close()
