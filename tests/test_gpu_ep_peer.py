"""The direct peer-store exchange of expert parallelism (include/moeinf.h: moeinf_ep_peer_*, csrc/ep_peer.h) with REAL
cross-rank traffic inside one process: two (three) engines = ranks on GPU 0, each driven by its own host thread on its own
stream — like the one-process-per-GPU product, minus hipIpc (ranks inside one process map each other's windows by pointer;
the process test, tests/test_gpu_ep_processes.py, covers the IPC mapping).  Every rank stores its routed rows straight into
the owners' windows, the owners' FFN stage 2 stores the outputs straight into the home ranks' windows, consumers wait on
flag words — no collective, no host staging.  Each rank's output must equal the oracle block of its own tokens.

Replaces in the reference: cudaDeviceEnablePeerAccess + implicit P2P `tensor.to(device)` row copies
(core/prefetch/archer_prefetch_handle.cpp:37-61, core/parallel/expert_dispatcher.cpp:284,405)."""
import os
import threading

import pytest
import torch

from helpers import R, acts, assert_block_close, make_weights

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _engines(family, world, h, f, e, k, n_shared, L, max_tokens, seed):
    from moe_infinity_amd import MoEEngine
    from moe_infinity_amd import config as Cf

    ws = [make_weights(family, h, f, e, seed + 10 * l, torch.bfloat16, n_shared=n_shared) for l in range(L)]
    et, rk = (Cf.EXPERT_MIXTRAL, Cf.ROUTER_MIXTRAL) if family == "mixtral" else (Cf.EXPERT_DEEPSEEK, Cf.ROUTER_DEEPSEEK)
    engs = []
    for r in range(world):
        eng = MoEEngine(Cf.EngineConfig(num_layers=L, num_experts=e, expert_type=et, hidden=h, inter=f, top_k=k, router_kind=rk,
                                        dtype=Cf.DTYPE_BF16, shared_inter=f * n_shared, device_memory_ratio=0.25, max_tokens=max_tokens,
                                        ep_rank=r, ep_size=world))
        for l in range(L):
            for i, ex in enumerate(ws[l][1]):
                if i % world == r:
                    eng.register_expert(l, i, ex)
            if ws[l][2]:
                eng.register_shared(l, ws[l][2])
        engs.append(eng)
    return ws, engs


_STREAMS = []  # one compute stream per rank, shared by every test of this file: each stream that has ever launched work holds
#                a hardware queue, and ranks that wait for each other ON the GPU must not share one (tests/conftest.py)


def _run_ranks(world, fn):
    """fn(rank) on one host thread per rank (ctypes releases the GIL inside the engine); re-raises the first failure"""
    errs = [None] * world
    while len(_STREAMS) < world:
        _STREAMS.append(torch.cuda.Stream(device=DEV, priority=-1))

    def body(r):
        try:
            torch.cuda.set_device(0)
            with torch.cuda.stream(_STREAMS[r]):
                fn(r)
                torch.cuda.current_stream().synchronize()
        except BaseException as ex:  # noqa: BLE001
            errs[r] = ex

    ths = [threading.Thread(target=body, args=(r,), daemon=True) for r in range(world)]
    for t in ths:
        t.start()
    for t in ths:
        t.join(timeout=120)
    assert not any(t.is_alive() for t in ths), "a rank did not finish: the ranks are waiting for each other (a poll of the exchange ran into its timeout?)"
    for ex in errs:
        if ex is not None:
            raise ex


@pytest.mark.parametrize("poll", [1, 0], ids=["consumers_poll_in_kernel", "one_wave_wait_kernels"])
@pytest.mark.parametrize("family,world", [("mixtral", 2), ("deepseek", 2), ("mixtral", 3)])
def test_peer_store_exchange_between_engines_of_one_process(family, world, poll, monkeypatch):
    monkeypatch.setenv("MOEINF_EP_PEER_POLL", str(poll))
    monkeypatch.setenv("MOEINF_EP_PEER_TIMEOUT_MS", "4000")
    e, k, n_shared = (8, 2, 0) if family == "mixtral" else (16, 4, 2)
    h, f, L, cap_tokens = 1024, 512, 2, 40
    per_rank = min(k, -(-e // world))
    ws, engs = _engines(family, world, h, f, e, k, n_shared, L, max_tokens=world * cap_tokens * per_rank, seed=5100)
    blobs = [eng.ep_peer_export(cap_tokens) for eng in engs]
    assert all(len(b) == 192 for b in blobs)
    for eng in engs:
        eng.ep_peer_attach(b"".join(blobs))
        t = eng.ep_transport()
        assert t["transport"] == "peer-store" and not t["shared_device"] and t["poll_in_kernels"] == bool(poll), t
    # token counts per call (every rank makes the same calls; its own count may differ): batch 1 (fused pack, self-indexing
    # owner), ragged small batches, 20 tokens (more rows than the self-indexing owner takes: wait + generic kernels + push;
    # pack as a launch of its own: > 64 KiB of rows), 40 tokens (> 64 pairs: indexed pack kernel)
    calls = [lambda r: 1, lambda r: 1, lambda r: 3 + r, lambda r: 20 - r, lambda r: 40 - 2 * r, lambda r: 2]
    outs = [[] for _ in range(world)]
    xs = [[] for _ in range(world)]
    oks = [False] * world

    def rank_body(r):
        oks[r] = engs[r].ep_peer_selftest()
        g = [w[0].to(DEV) for w in ws]
        for ci, tf in enumerate(calls):
            for l in range(L):
                x = acts(tf(r), h, torch.bfloat16, 5200 + 31 * ci + 7 * l + 1000 * r)
                xd = x.to(DEV)
                out = torch.empty_like(xd)
                engs[r].ep_moe_forward(l, xd, g[l], out)
                xs[r].append((l, x))
                outs[r].append(out)

    _run_ranks(world, rank_body)
    assert all(oks), f"self-test: {oks}"
    for r in range(world):
        for (l, x), out in zip(xs[r], outs[r]):
            if family == "mixtral":
                ref = R.block_mixtral(x[None], ws[l][0], ws[l][1], top_k=k)
            else:
                ref = R.block_deepseek(x[None], ws[l][0], ws[l][1], k, shared=ws[l][2])
            assert_block_close(out.cpu(), ref, torch.bfloat16, f"peer-store, {family}, rank {r} of {world}, layer {l}, {x.shape[0]} tokens")
        assert engs[r].ep_transport()["exchanges"] == 1 + len(calls) * L
    for eng in engs:
        eng.sync()  # reads the device error flag: a poll that timed out would be reported here
    for eng in engs:
        eng.close()


def test_peer_store_bootstrap_errors_are_local_and_never_block():
    """every bootstrap step fails locally with a message (wrong blob size, blobs of another shape, attach before export);
    a self-test whose peer never shows up returns False after the timeout instead of hanging"""
    from moe_infinity_amd._lib import MoeInfError

    os.environ["MOEINF_EP_PEER_TIMEOUT_MS"] = "300"
    try:
        ws, engs = _engines("mixtral", 2, 256, 256, 8, 2, 0, 1, max_tokens=64, seed=5300)
        with pytest.raises(MoeInfError):
            engs[0].ep_peer_attach(bytes(2 * 192))  # before export
        b0 = engs[0].ep_peer_export(8)
        b1 = engs[1].ep_peer_export(16)  # another capacity: a different window
        with pytest.raises(MoeInfError, match="different window"):
            engs[0].ep_peer_attach(b0 + b1)
        with pytest.raises(MoeInfError):
            engs[0].ep_peer_attach(b0)  # one blob short
        with pytest.raises(MoeInfError):
            engs[0].ep_peer_attach(b1 + b0)  # wrong order
        engs[1].close()
        ws, e1 = _engines("mixtral", 2, 256, 256, 8, 2, 0, 1, max_tokens=64, seed=5300)
        e1[0].close()
        b1 = e1[1].ep_peer_export(8)
        engs[0].ep_peer_attach(b0 + b1)
        e1[1].ep_peer_attach(b0 + b1)
        assert engs[0].ep_peer_selftest() is False  # rank 1 never runs its half: bounded wait, no hang
        engs[0].close()
        e1[1].close()
    finally:
        os.environ.pop("MOEINF_EP_PEER_TIMEOUT_MS", None)
