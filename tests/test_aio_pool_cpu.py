"""The two-priority block reader of the disk tier (csrc/aio_pool.h) through the C ABI (moeinf_aio_*): data integrity,
high-before-low service, promotion, error paths.  Host-only; restates the queue discipline of the reference's
ArcherPrioAioContext::Schedule (core/aio/archer_prio_aio_handle.cpp:123-169)."""
import ctypes as C
import os

import numpy as np
import pytest

from moe_infinity_amd import MoeInfError, load_library
from moe_infinity_amd._lib import check


@pytest.fixture(scope="module")
def lib():
    return load_library()


def _aligned(nbytes, align=4096):
    raw = np.empty(nbytes + align, np.uint8)
    off = (-raw.ctypes.data) % align
    return raw[off:off + nbytes]


def _pool(lib, threads, block):
    h = C.c_void_p()
    check(lib.moeinf_aio_create(threads, block, C.byref(h)))
    return h


def _submit(lib, pool, path, dst, nbytes, offset, high, direct=True):
    rid = C.c_int64(0)
    check(lib.moeinf_aio_submit_read(pool, path.encode(), dst.ctypes.data_as(C.c_void_p), nbytes, offset, int(high), int(direct), C.byref(rid)))
    return rid.value


def _stats(lib, pool):
    out = (C.c_int64 * 5)()
    check(lib.moeinf_aio_stats(pool, out))
    return dict(zip(("blocks_high", "blocks_low", "bytes", "promoted", "direct_fallbacks"), list(out)))


def test_reads_are_exact_at_every_geometry(lib, tmp_path):
    rng = np.random.default_rng(0)
    data = rng.integers(0, 256, size=5 * 4096 * 37 + 1234, dtype=np.uint8)  # not a multiple of the alignment
    # the offload files are padded to the alignment: do the same, the payload ends inside the padding
    padded = np.concatenate([data, np.zeros((-data.size) % 4096, np.uint8)])
    f = tmp_path / "archer_param_0"
    padded.tofile(f)
    pool = _pool(lib, 3, 64 * 1024)
    reqs = []
    for off, n, direct in ((0, data.size, True), (4096 * 5, 4096 * 9, True), (4096 * 3, 100_001, True), (777, 50_000, False),
                           (4096 * 40, data.size - 4096 * 40, True), (0, 1, False)):
        dst = _aligned((n + 4095) // 4096 * 4096)
        dst[:] = 0xEE
        reqs.append((_submit(lib, pool, str(f), dst, n, off, high=False, direct=direct), dst, off, n))
    for rid, dst, off, n in reqs:
        check(lib.moeinf_aio_wait(pool, rid))
        assert np.array_equal(dst[:n], data[off:off + n]), (off, n)
    st = _stats(lib, pool)
    assert st["bytes"] >= sum(n for _, _, _, n in reqs) and st["blocks_low"] >= 6 and st["blocks_high"] == 0
    check(lib.moeinf_aio_destroy(pool))


def test_high_priority_overtakes_queued_low_priority_blocks(lib, tmp_path):
    """One worker, many queued speculative blocks, then an on-demand read: it must finish while most of the speculative
    blocks are still waiting (the reference lets exactly one low block through per scheduler turn)."""
    blk = 64 * 1024
    n_low, low_blocks = 6, 48
    f = tmp_path / "p"
    np.arange(low_blocks * blk // 8, dtype=np.int64).tofile(f)
    size = low_blocks * blk
    pool = _pool(lib, 1, blk)
    lows = []
    for _ in range(n_low):
        dst = _aligned(size)
        lows.append((_submit(lib, pool, str(f), dst, size, 0, high=False), dst))
    hdst = _aligned(4 * blk)
    hid = _submit(lib, pool, str(f), hdst, 4 * blk, 8 * blk, high=True)
    check(lib.moeinf_aio_wait(pool, hid))
    st = _stats(lib, pool)
    assert st["blocks_high"] == 4
    assert st["blocks_low"] < n_low * low_blocks // 2, f"the demand read waited behind {st['blocks_low']} speculative blocks"
    assert np.array_equal(hdst.view(np.int64), np.arange(8 * blk // 8, 12 * blk // 8, dtype=np.int64))
    # promote the LAST speculative request: it finishes before the ones queued ahead of it
    check(lib.moeinf_aio_promote(pool, lows[-1][0]))
    check(lib.moeinf_aio_wait(pool, lows[-1][0]))
    done = C.c_int32(0)
    check(lib.moeinf_aio_done(pool, lows[-2][0], C.byref(done)))
    assert done.value == 0, "a promoted request must not wait for the low-priority requests queued before it"
    assert _stats(lib, pool)["promoted"] == 1
    for rid, dst in lows[:-1]:
        check(lib.moeinf_aio_wait(pool, rid))
        assert np.array_equal(dst.view(np.int64), np.arange(size // 8, dtype=np.int64))
    check(lib.moeinf_aio_destroy(pool))


def test_errors_are_reported_not_fatal(lib, tmp_path):
    pool = _pool(lib, 2, 4096)
    dst = _aligned(8192)
    rid = _submit(lib, pool, str(tmp_path / "missing"), dst, 8192, 0, high=True)
    with pytest.raises(MoeInfError, match="open"):
        check(lib.moeinf_aio_wait(pool, rid))
    f = tmp_path / "short"
    np.zeros(4096, np.uint8).tofile(f)
    rid = _submit(lib, pool, str(f), dst, 8192, 0, high=False, direct=False)
    with pytest.raises(MoeInfError, match="shorter"):
        check(lib.moeinf_aio_wait(pool, rid))
    with pytest.raises(MoeInfError, match="unknown"):
        check(lib.moeinf_aio_wait(pool, 12345))
    check(lib.moeinf_aio_destroy(pool))
