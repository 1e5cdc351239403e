"""The direct peer-store exchange of expert parallelism (include/moeinf.h: moeinf_ep_peer_*, csrc/ep_peer.h) inside one
process: every kernel path of the exchange at world size 1 (the rank stores into its own window), and the bootstrap's error
behaviour.  Ranks that wait for each other ON the GPU need hardware queues of their own — guaranteed between processes, not
between streams of one process (the HIP runtime shares its queues between streams: a waiting kernel can sit in front of the
kernel it waits for) — so the multi-rank runs, with hipIpc-mapped windows and both consumer modes, are
tests/test_gpu_ep_processes.py.

Replaces in the reference: cudaDeviceEnablePeerAccess + implicit P2P `tensor.to(device)` row copies
(core/prefetch/archer_prefetch_handle.cpp:37-61, core/parallel/expert_dispatcher.cpp:284,405)."""
import os

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


def test_peer_store_at_world_size_one_takes_every_owner_path():
    """One rank storing into its own window (no second stream, no second process: nothing here can wait for anything that is not
    already enqueued in front of it): fused and separate pack launches, the self-indexing owner kernels (polls inside the
    kernels) and the generic owner path (wait kernel + index + FFN + push kernel), both consumer modes — each against the oracle
    block.  The multi-rank runs are tests/test_gpu_ep_processes.py: real processes, hipIpc-mapped windows."""
    for family, e, k, n_shared in (("mixtral", 8, 2, 0), ("deepseek", 16, 4, 2)):
        for poll in ("1", "0"):
            os.environ["MOEINF_EP_PEER_POLL"] = poll
            os.environ["MOEINF_EP_PEER_TIMEOUT_MS"] = "2000"
            try:
                h, f, L = 1024, 512, 2
                ws, engs = _engines(family, 1, h, f, e, k, n_shared, L, max_tokens=40 * k, seed=5100)
                eng = engs[0]
                for cap_tokens, calls in ((8, (1, 1, 3, 8)), ):  # cap 8 tokens -> at most 8*k <= 64 received rows: the self-indexing kernels
                    eng.ep_peer_attach(eng.ep_peer_export(cap_tokens))
                    assert eng.ep_peer_selftest()
                    t = eng.ep_transport()
                    assert t["transport"] == "peer-store" and t["poll_in_kernels"] == (poll == "1"), t
                    for ci, tok in enumerate(calls):
                        for l in range(L):
                            x = acts(tok, h, torch.bfloat16, 5200 + 31 * ci + 7 * l)
                            out = torch.empty(tok, h, dtype=torch.bfloat16, device=DEV)
                            for _ in range(2):  # decision path first, then the sync-free path
                                eng.ep_moe_forward(l, x.to(DEV), ws[l][0].to(DEV), out)
                            eng.sync()
                            ref = (R.block_mixtral(x[None], ws[l][0], ws[l][1], top_k=k) if family == "mixtral"
                                   else R.block_deepseek(x[None], ws[l][0], ws[l][1], k, shared=ws[l][2]))
                            assert_block_close(out.cpu(), ref, torch.bfloat16, f"peer-store world 1, {family}, poll={poll}, layer {l}, {tok} tokens")
                eng.close()
                # a window for 40 tokens: 40*k > 64 rows -> the generic owner path; 40 tokens x K = 80 pairs -> the indexed pack kernel
                ws, engs = _engines(family, 1, h, f, e, k, n_shared, L, max_tokens=40 * k, seed=5100)
                eng = engs[0]
                eng.ep_peer_attach(eng.ep_peer_export(40))
                assert eng.ep_peer_selftest()
                for ci, tok in enumerate((1, 20, 40)):
                    for l in range(L):
                        x = acts(tok, h, torch.bfloat16, 5300 + 31 * ci + 7 * l)
                        out = torch.empty(tok, h, dtype=torch.bfloat16, device=DEV)
                        eng.ep_moe_forward(l, x.to(DEV), ws[l][0].to(DEV), out)
                        eng.sync()
                        ref = (R.block_mixtral(x[None], ws[l][0], ws[l][1], top_k=k) if family == "mixtral"
                               else R.block_deepseek(x[None], ws[l][0], ws[l][1], k, shared=ws[l][2]))
                        assert_block_close(out.cpu(), ref, torch.bfloat16, f"peer-store world 1 (generic owner path), {family}, poll={poll}, layer {l}, {tok} tokens")
                eng.close()
            finally:
                os.environ.pop("MOEINF_EP_PEER_POLL", None)
                os.environ.pop("MOEINF_EP_PEER_TIMEOUT_MS", None)


@pytest.mark.parametrize("family,e,k,n_shared", [("mixtral", 8, 2, 0), ("deepseek", 16, 4, 2), ("deepseek", 64, 6, 2)])
def test_batch1_broadcast_form_at_world_size_one(family, e, k, n_shared):
    """One token per rank with the caller's uniform-token promise: the home rank broadcasts (row, gate logits) and the owner's FFN
    stage 1 routes for itself (four launches per layer).  First forward of a layer: its experts are not resident yet — the
    slow path (broadcast launch + unpack into the routed form + generic owner kernels); from the second on: the fast path.
    Both against the oracle block, bit-stable, and identical to what the routed form returns."""
    os.environ["MOEINF_EP_PEER_TIMEOUT_MS"] = "2000"
    try:
        h, f, L = 512, 384, 2
        ws, engs = _engines(family, 1, h, f, e, k, n_shared, L, max_tokens=16, seed=5400)
        eng = engs[0]
        eng.ep_peer_attach(eng.ep_peer_export(4))
        assert eng.ep_peer_selftest()
        for seed in range(3):
            for l in range(L):
                x = acts(1, h, torch.bfloat16, 5500 + 10 * seed + l)
                ref = (R.block_mixtral(x[None], ws[l][0], ws[l][1], top_k=k) if family == "mixtral"
                       else R.block_deepseek(x[None], ws[l][0], ws[l][1], k, shared=ws[l][2]))
                eng.ep_set_uniform_tokens(False)
                routed = torch.empty(1, h, dtype=torch.bfloat16, device=DEV)
                eng.ep_moe_forward(l, x.to(DEV), ws[l][0].to(DEV), routed)
                eng.ep_set_uniform_tokens(True)
                outs = []
                for _ in range(2):
                    out = torch.empty(1, h, dtype=torch.bfloat16, device=DEV)
                    eng.ep_moe_forward(l, x.to(DEV), ws[l][0].to(DEV), out)
                    eng.sync()
                    outs.append(out)
                for out in outs + [routed]:
                    assert_block_close(out.cpu(), ref, torch.bfloat16, f"broadcast form, {family} e={e}, layer {l}")
                assert torch.equal(outs[0], outs[1]), "a repeated forward must be bit-identical"
                if not n_shared:  # (a hidden shared expert's stage 2 splits its reduction over 4 waves here and 8 in the routed
                    #               form's router launch: the same numbers in another fp32 summation order)
                    assert torch.equal(outs[0], routed), "broadcast and routed forms must agree bit for bit"
        st = eng.stats()
        assert st["expert_misses"] <= L * e
        eng.close()
    finally:
        os.environ.pop("MOEINF_EP_PEER_TIMEOUT_MS", None)


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


def test_a_released_window_starts_over_at_exchange_zero_and_the_timeout_getter_round_trips():
    """moeinf_ep_peer_release kept the exchange number, so a group whose ranks had fallen out of step (one of
    them failed before it took its number) met again at DIFFERENT epochs after the next export / attach and failed at once
    with flag 2 or 3.  Release resets it; and a host layer that shortens the exchange timeout for a probation restores the
    value that was in force (getter), not the environment's default."""
    os.environ["MOEINF_EP_PEER_TIMEOUT_MS"] = "2000"
    try:
        h, f, e, k, L = 512, 256, 8, 2, 1
        ws, engs = _engines("mixtral", 1, h, f, e, k, 0, L, max_tokens=32, seed=5400)
        eng = engs[0]
        gate = ws[0][0].to(DEV)
        eng.ep_peer_attach(eng.ep_peer_export(8))
        assert eng.ep_peer_selftest()
        for i in range(3):  # three exchanges: the window's number moves on
            x = acts(2, h, torch.bfloat16, 5410 + i)
            out = torch.empty(2, h, dtype=torch.bfloat16, device=DEV)
            eng.ep_moe_forward(0, x.to(DEV), gate, out)
            eng.sync()
        n0 = eng.ep_transport()["exchanges"]
        assert n0 >= 3
        assert eng.ep_peer_get_timeout_ms() == 2000
        assert eng.ep_peer_set_timeout_ms(750) == 2000 and eng.ep_peer_get_timeout_ms() == 750
        eng.ep_peer_release()
        assert eng.ep_transport()["transport"] != "peer-store"
        eng.ep_peer_attach(eng.ep_peer_export(8))
        assert eng.ep_transport()["exchanges"] == 0, "a fresh window counts its exchanges from zero"
        assert eng.ep_peer_selftest()
        x = acts(2, h, torch.bfloat16, 5420)
        eng.ep_moe_forward(0, x.to(DEV), gate, out)
        eng.sync()
        assert_block_close(out.cpu(), R.block_mixtral(x[None], ws[0][0], ws[0][1], top_k=k), torch.bfloat16, "forward over the re-exported window")
        eng.close()
    finally:
        os.environ.pop("MOEINF_EP_PEER_TIMEOUT_MS", None)
