"""The host-side mirrors of the reference's Python interface (dispatcher shim, dispatch_local, the
Sync* blocks, tracer/predictor/prefetcher) on the HIP engine, checked against the oracle.  -m gpu."""
import types

import numpy as np
import pytest
import torch

from helpers import R, acts, assert_block_close, assert_model_close, engine_for, make_weights, register_all

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def test_dispatch_local_like_the_reference_blocks_use_it():
    """Python router + masks (oracle, standing in for the reference block's torch code) -> our
    DistributedExpertExecutor.dispatch_local -> list[(out, layer, expert, hit)] -> Python combine."""
    from moe_infinity_amd.expert_executor import DistributedExpertExecutor, ExpertDispatcher

    h, f, e, k, t = 256, 512, 8, 2, 12
    gate, experts, _ = make_weights("mixtral", h, f, e, 500, torch.bfloat16)
    eng = engine_for("mixtral", h, f, e, k, torch.bfloat16, max_tokens=t)
    disp = ExpertDispatcher(eng)
    for i, ex in enumerate(experts):
        disp.register_expert(0, i, ex)
    ex_ = DistributedExpertExecutor(None)
    ex_.set_expert_dispatcher(disp)
    x = acts(t, h, torch.bfloat16, 501)
    sel, w, _ = R.route_mixtral(x, gate, k)
    router_mask, weights_mask = R.masks_from_topk(sel, w, e)
    ref = R.block_mixtral(x[None], gate, experts, top_k=k)
    for call in range(2):
        res = ex_.dispatch_local(x.to(DEV), router_mask.to(DEV), 0)
        assert [r[2] for r in res] == sorted(ref.expert_out)  # ascending expert order
        final = torch.zeros(t, h, dtype=torch.bfloat16)
        for out, layer, idx, hit in res:
            assert layer == 0 and hit == call  # first call: every expert fetched on demand; second: all resident
            assert_model_close(out, ref.expert_out[idx], torch.bfloat16, f"expert {idx} output")
            tok = router_mask[:, idx]
            final[tok, :] += out.cpu() * weights_mask[tok, idx][:, None]  # mixtral.py:96-101
        assert_block_close(final, ref, torch.bfloat16, "python-combined block output")
    eng.close()


def test_sync_mixtral_block_returns_what_the_reference_block_returns():
    from moe_infinity_amd import MoEEngine
    from moe_infinity_amd.blocks import SyncMixtralSparseMoeBlock

    h, f, e, k = 256, 512, 8, 2
    cfg = types.SimpleNamespace(hidden_size=h, intermediate_size=f, num_local_experts=e, num_experts_per_tok=k)
    gate, experts, _ = make_weights("mixtral", h, f, e, 510, torch.bfloat16)
    blk = SyncMixtralSparseMoeBlock(cfg).to(torch.bfloat16).to(DEV)
    with torch.no_grad():
        blk.gate.weight.copy_(gate)
    eng = MoEEngine(SyncMixtralSparseMoeBlock.engine_config(cfg, num_layers=1, device_memory_ratio=0.25, max_tokens=16))
    blk.attach_engine(eng, 0)
    blk.register_experts(experts)
    x = acts(10, h, torch.bfloat16, 511).reshape(2, 5, h)
    out, router_logits = blk(x.to(DEV))
    ref = R.block_mixtral(x, gate, experts, top_k=k)
    assert out.shape == x.shape and router_logits.shape == (10, e)
    assert torch.equal(router_logits.float().cpu(), ref.logits.float())
    assert_block_close(out, ref, torch.bfloat16, "block output")
    eng.close()


def test_deepseek_and_switch_and_nllb_blocks():
    from moe_infinity_amd import MoEEngine
    from moe_infinity_amd.blocks import DeepseekMoEBlock, SyncNllbMoeSparseMLP, SyncSwitchTransformersSparseMLP

    h, f = 256, 192
    # deepseek
    e, k = 16, 4
    cfg = types.SimpleNamespace(hidden_size=h, moe_intermediate_size=f, n_routed_experts=e, num_experts_per_tok=k,
                                n_shared_experts=2, norm_topk_prob=False, routed_scaling_factor=1.0, topk_method="greedy",
                                n_group=None, topk_group=None)
    gate, experts, shared = make_weights("deepseek", h, f, e, 520, torch.bfloat16, n_shared=2)
    blk = DeepseekMoEBlock(cfg).to(torch.bfloat16).to(DEV)
    with torch.no_grad():
        blk.gate.weight.copy_(gate)
    eng = MoEEngine(DeepseekMoEBlock.engine_config(cfg, 1, device_memory_ratio=0.25, max_tokens=8))
    blk.attach_engine(eng, 0)
    blk.register_experts(experts, shared)
    x = acts(6, h, torch.bfloat16, 521).reshape(1, 6, h)
    assert_block_close(blk(x.to(DEV)), R.block_deepseek(x, gate, experts, k, shared=shared), torch.bfloat16, "deepseek block")
    eng.close()
    # switch (fp32)
    e = 8
    cfg = types.SimpleNamespace(d_model=h, d_ff=f, num_experts=e, expert_capacity=3)
    gate, experts, _ = make_weights("switch", h, f, e, 530, torch.float32, gate_std=0.5)
    blk = SyncSwitchTransformersSparseMLP(cfg).to(DEV)
    with torch.no_grad():
        blk.router.classifier.weight.copy_(gate)
    eng = MoEEngine(SyncSwitchTransformersSparseMLP.engine_config(cfg, 1, dtype=1, device_memory_ratio=0.25, max_tokens=32))
    blk.attach_engine(eng, 0)
    blk.register_experts(experts)
    x = acts(24, h, torch.float32, 531).reshape(2, 12, h)
    out, (logits, expert_index) = blk(x.to(DEV))
    ref = R.block_switch(x, gate, experts, expert_capacity=3)
    assert logits.shape == (2, 12, e) and expert_index.shape == (2, 12)
    assert int((ref.router_mask.sum(-1) == 0).sum()) > 0, "the case must exercise capacity drops"
    assert torch.equal(expert_index.cpu().reshape(-1), ref.topk_idx.reshape(-1)), "argmax(router_mask): dropped tokens report 0"
    assert_block_close(out, ref, torch.float32, "switch block")
    eng.close()
    # nllb
    e = 16
    cfg = types.SimpleNamespace(d_model=h, num_experts=e, normalize_router_prob_before_dropping=False)
    gate, experts, _ = make_weights("nllb", h, f, e, 540, torch.bfloat16, gate_std=0.5)
    blk = SyncNllbMoeSparseMLP(cfg, f).to(torch.bfloat16).to(DEV)
    with torch.no_grad():
        blk.router.classifier.weight.copy_(gate)
    eng = MoEEngine(SyncNllbMoeSparseMLP.engine_config(cfg, f, 1, device_memory_ratio=0.25, max_tokens=16))
    blk.attach_engine(eng, 0)
    blk.register_experts(experts)
    x = acts(9, h, torch.bfloat16, 541).reshape(3, 3, h)
    out, (router_probs, top1) = blk(x.to(DEV))
    ref = R.block_nllb(x, gate, experts)
    assert torch.equal(router_probs.bool().cpu(), ref.weights_mask.bool())
    assert torch.equal(top1.cpu(), torch.argmax(ref.extra["top_1_mask"], dim=-1))
    assert_block_close(out, ref, torch.bfloat16, "nllb block")
    eng.close()


def test_predictor_and_prefetcher_drive_the_engine():
    """Activation-aware prefetch revived (mixtral.py:71-85 is commented out in the reference): with a
    history that matches the sequence, predicted experts are copied ahead of use and arrive as hits;
    numerics are unchanged."""
    from moe_infinity_amd import MoEEngine
    from moe_infinity_amd import config as Cf
    from moe_infinity_amd.blocks import SyncMixtralSparseMoeBlock
    from moe_infinity_amd.memory import ExpertPredictor, ExpertPrefetcher, ExpertTracer

    h, f, e, k, L, t = 256, 512, 8, 2, 4, 1
    cfg = types.SimpleNamespace(hidden_size=h, intermediate_size=f, num_local_experts=e, num_experts_per_tok=k)
    ws = [make_weights("mixtral", h, f, e, 600 + l, torch.bfloat16) for l in range(L)]
    slot = 3 * f * h * 2
    eng = MoEEngine(SyncMixtralSparseMoeBlock.engine_config(cfg, L, device_memory_bytes=12 * slot, max_tokens=t))
    blocks = []
    for l in range(L):
        b = SyncMixtralSparseMoeBlock(cfg).to(torch.bfloat16).to(DEV)
        with torch.no_grad():
            b.gate.weight.copy_(ws[l][0])
        b.attach_engine(eng, l)
        b.register_experts(ws[l][1])
        blocks.append(b)
    xs = [[acts(t, h, torch.bfloat16, 7000 + 10 * s + l) for l in range(L)] for s in range(6)]
    # history = the exact EAM this input sequence produces (oracle routing), so the nearest EAM is perfect
    eam = np.zeros((1, L, e), np.float32)
    for s in range(6):
        for l in range(L):
            sel, _, _ = R.route_mixtral(xs[s][l], ws[l][0], k)
            for i in sel.reshape(-1).tolist():
                eam[0, l, i] += 1
    tracer = ExpertTracer(4, L, e)
    tracer.load_trace(np.repeat(eam, 4, axis=0))
    pred = ExpertPredictor(L, e)
    pred.add_tracer(tracer)
    pf = ExpertPrefetcher(L, e, tracer)
    pf.set_archer_engine(eng)
    seq = tracer.create_entry()
    for b in blocks:
        b.expert_predictor, b.expert_prefetcher, b.seq_id_list = pred, pf, [seq]
    for s in range(6):
        for l in range(L):
            out, _ = blocks[l](xs[s][l].to(DEV)[None])
            ref = R.block_mixtral(xs[s][l][None], ws[l][0], ws[l][1], top_k=k)
            assert_block_close(out, ref, torch.bfloat16, f"step {s} layer {l}")
    st = eng.stats()
    assert st["prefetch_issued"] > 0 and st["prefetch_useful"] > 0
    assert st["expert_hits"] + st["expert_misses"] == 6 * L * k  # every dispatch accounted for exactly once
    eng.close()


def test_experts_loaded_from_a_reference_format_offload_directory(tmp_path):
    """disk tier -> pinned arena -> HBM: experts written with prefetch_handle.offload() semantics into an
    archer_index/archer_param directory are registered by tensor id (blob order) and give the same block
    output as the in-memory registration."""
    from moe_infinity_amd.offload_store import OffloadStore

    h, f, e, k, t = 256, 512, 8, 2, 6
    gate, experts, _ = make_weights("mixtral", h, f, e, 700, torch.bfloat16)
    st = OffloadStore(str(tmp_path))
    ids, tid = {}, 10
    for i, ex in enumerate(experts):  # tensor ids in named_parameters order: w1, w2, w3 (model_offload.py:645-671)
        ids[i] = []
        for w in ex:
            st.offload(w, tid)
            ids[i].append(tid)
            tid += 1
    st.close()
    st = OffloadStore(str(tmp_path))  # reopen: the index is parsed from disk
    eng = engine_for("mixtral", h, f, e, k, torch.bfloat16, max_tokens=t)
    for i in range(e):
        st.register_expert(eng, 0, i, ids[i])
    x = acts(t, h, torch.bfloat16, 701)
    out = eng.forward(0, x.to(DEV), gate.to(DEV))
    ref = R.block_mixtral(x[None], gate, experts, top_k=k)
    assert_block_close(out, ref, torch.bfloat16, "block output from a disk-loaded model")
    from moe_infinity_amd import MoeInfError
    with pytest.raises(MoeInfError):
        st.register_expert(eng, 0, 0, [ids[0][0], 9999, ids[0][2]])  # unknown tensor id
    eng.close()
    st.close()


def test_expert_parallel_module_through_rccl_world_size_1():
    """moe_infinity_amd.ep.ExpertParallelMoE with the HIP ops and the real collective (torch.distributed "nccl" =
    RCCL) at world size 1: route -> pack -> all_to_all -> owner FFN -> all_to_all -> combine must equal the oracle
    block, for a decode-sized and a 40-token batch, Mixtral and DeepSeek (shared expert on the home rank)."""
    import os
    import socket

    import torch.distributed as dist

    from moe_infinity_amd import MoEEngine
    from moe_infinity_amd import config as Cf
    from moe_infinity_amd.ep import ExpertParallelMoE, HipEpOps

    created = False
    if not dist.is_initialized():
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            port = s.getsockname()[1]
        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(port)
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device(DEV))
        created = True
    try:
        for family in ("mixtral", "deepseek"):
            h, f, e, k, n_shared = 256, 192, 8, 2, 0
            if family == "deepseek":
                e, k, n_shared = 16, 4, 2
            gate, experts, shared = make_weights(family, h, f, e, 540, torch.bfloat16, n_shared=n_shared)
            tmax = 40
            eng = engine_for(family, h, f, e, k, torch.bfloat16, n_shared=n_shared, max_tokens=tmax)
            register_all(eng, experts, shared)
            # torch.distributed transport (five host calls per layer), then the engine's own RCCL communicator (one host call
            # per layer: moeinf_ep_moe_forward) — bootstrapped through the process group and self-tested before use
            # and the direct peer-store exchange (no collective; at world size 1 the rank stores into its own window)
            for transport in ("torch", "rccl", "peer-store", "auto"):
                ep = ExpertParallelMoE(HipEpOps(eng), h, k, tmax, torch.bfloat16, DEV, num_experts=e, transport=transport)
                assert ep.transport == ("peer-store" if transport == "auto" else transport), f"{transport}: {ep.native_note}"
                assert ep.native == (transport != "torch")
                if ep.native:
                    assert eng.ep_transport()["transport"] == ep.transport
                g = gate.to(DEV)
                for t in (1, 3, 40):
                    x = acts(t, h, torch.bfloat16, 541 + t)
                    for _ in range(3):  # the first forward takes the decision path, the others the sync-free (self-indexing) one
                        out = ep.forward(0, x.to(DEV), g)
                    if family == "mixtral":
                        ref = R.block_mixtral(x[None], gate, experts, top_k=k)
                    else:
                        ref = R.block_deepseek(x[None], gate, experts, k, shared=shared)
                    assert_block_close(out, ref, torch.bfloat16, f"EP module, {family}, {t} tokens, transport={ep.transport}")
                if ep.native:
                    ep.profile = True
                    ep.forward(0, x.to(DEV), g)
                    ph = ep.phase_times_us()
                    assert ph["calls"] == 1 and all(ph[p] > 0 for p in ep.PHASES), ph
                    ep.profile = False
            eng.close()
    finally:
        if created:
            dist.destroy_process_group()


@pytest.mark.parametrize("family", ["mixtral", "deepseek", "switch", "nllb"])
def test_standalone_combine_export_matches_the_fused_forward(family):
    """moeinf_combine (the decomposed `combine(...)` for callers that keep the Python router):
    expert outputs from moeinf_dispatch_mask + the router's top-k (ids, weights) -> the block output, bit-identical to
    what moeinf_moe_forward produces for the same tokens (same summation order, same rounding points)."""
    h, f, e, t = 256, 512, 8, 23
    k = {"mixtral": 2, "deepseek": 3, "switch": 1, "nllb": 2}[family]
    dtype = torch.float32 if family == "switch" else torch.bfloat16
    gate, experts, _ = make_weights(family, h, f, e, 880, dtype, gate_std=0.5 if family in ("switch", "nllb") else 0.02)
    kw = dict(expert_capacity=4) if family == "switch" else {}
    eng = engine_for(family, h, f, e, k, dtype, max_tokens=t, **kw)
    register_all(eng, experts)
    x = acts(t, h, dtype, 881).to(DEV)
    g = gate.to(DEV)
    batch_rows = 1
    full = eng.forward(0, x, g, batch_rows=batch_rows).clone()
    r = eng.routing()
    idx = torch.from_numpy(r["topk_idx"]).to(torch.int32)          # -1 = dropped pair (Switch capacity, NLLB zero weight)
    w = torch.from_numpy(r["topk_w"]).to(torch.float32)
    mask = torch.zeros(t, e, dtype=torch.bool)
    for ti in range(t):
        for j in idx[ti]:
            if j >= 0:
                mask[ti, int(j)] = True
    y, counts, _ = eng.dispatch_mask(0, x, mask.to(DEV))
    assert int(counts.sum()) == int((idx >= 0).sum())
    prob = w[:, 0].contiguous().to(DEV) if family == "switch" else None
    out = eng.combine(y.contiguous(), idx.to(DEV), w.to(DEV), x2=x, router_prob=prob)
    assert torch.equal(out.cpu(), full.cpu()), f"{family}: standalone combine differs from the fused forward"
    eng.close()


def test_skewed_routing_on_the_sync_free_path():
    """Every token routed to the same two experts (200 rows each) while the sync-free path sizes its launch from the
    ESTIMATE min(T, 1.5*T*K/E + 1) = 76 rows: the first forward takes the decision path (exact counts), the second
    and third the sync-free path — all three must match the oracle."""
    h, f, e, k, t = 256, 512, 8, 2, 200
    gate, experts, _ = make_weights("mixtral", h, f, e, 910, torch.bfloat16)
    gate = gate.clone()
    gate[:, 0] = 0
    gate[0, 0], gate[1, 0] = 1.0, 0.9          # feature 0 decides: experts 0 and 1 win for every token
    x = acts(t, h, torch.bfloat16, 911).clone()
    x[:, 0] = 4.0
    eng = engine_for("mixtral", h, f, e, k, torch.bfloat16, max_tokens=t)
    register_all(eng, experts)
    ref = R.block_mixtral(x[None], gate, experts, top_k=k)
    assert set(ref.topk_idx.reshape(-1).tolist()) == {0, 1}
    for i in range(3):
        out = eng.forward(0, x.to(DEV), gate.to(DEV))
        if i == 0:
            eng.sync_copies()  # every expert resident and landed: the next forwards are sync-free
            for ee in range(e):
                if not eng.is_resident(0, ee):
                    eng.prefetch(0, [ee])
            eng.sync_copies()
        r = eng.routing()
        assert list(r["counts"]) == [t, t, 0, 0, 0, 0, 0, 0]
        assert_block_close(out, ref, torch.bfloat16, f"skewed routing, forward {i}")
    eng.close()


def test_engine_side_predictor_requests_the_next_layers_experts_without_host_code():
    """moeinf_set_predictor: the native tracer attached to the engine is fed from the routing mirrors and, on forwards
    that take the decision path, requests the predicted experts of the next layers itself (no eng.routing(), no Python
    predictor between the layers).  With a history that matches the sequence the copies arrive as hits; the EAM the
    engine accumulated equals the routing that actually happened; numerics are unchanged."""
    from moe_infinity_amd.engine import ExpertTracerNative

    h, f, e, k, L, t = 256, 512, 8, 2, 4, 1
    ws = [make_weights("mixtral", h, f, e, 650 + l, torch.bfloat16) for l in range(L)]
    slot = 3 * f * h * 2
    eng = engine_for("mixtral", h, f, e, k, torch.bfloat16, max_tokens=t, num_layers=L, device_memory_bytes=12 * slot)
    for l in range(L):
        register_all(eng, ws[l][1], layer=l)
    steps = 6
    # the same token every step: the routing repeats, so the history below predicts it exactly (and with 12 slots for 32
    # experts no layer is ever fully resident: every forward takes the decision path, where the predictor runs)
    xs = [[acts(t, h, torch.bfloat16, 7300 + l) for l in range(L)] for s in range(steps)]
    eam = np.zeros((1, L, e), np.float32)
    for s in range(steps):
        for l in range(L):
            sel, _, _ = R.route_mixtral(xs[s][l], ws[l][0], k)
            for i in sel.reshape(-1).tolist():
                eam[0, l, i] += 1
    tr = ExpertTracerNative(L, e, 4)
    tr.load_trace(np.repeat(eam, 4, axis=0))
    seq = tr.create_entry()
    eng.set_predictor(tr, seq, lookahead_layers=2, min_share=0.1, max_experts=8)
    gates = [w[0].to(DEV) for w in ws]
    for s in range(steps):
        for l in range(L):
            out = eng.forward(l, xs[s][l].to(DEV), gates[l])
            # a speculative copy is never started while an on-demand copy is on the link: let this layer's demand copies
            # land and pump the queue (any API call does), so that the test does not depend on how long the host dawdles
            torch.cuda.synchronize()
            eng.stats()
            ref = R.block_mixtral(xs[s][l][None], ws[l][0], ws[l][1], top_k=k)
            assert_block_close(out, ref, torch.bfloat16, f"step {s} layer {l}")
    st = eng.stats()
    assert st["prefetch_issued"] > 0 and st["prefetch_useful"] > 0, st
    assert st["expert_hits"] + st["expert_misses"] == steps * L * k
    torch.cuda.synchronize()
    eng.stats()  # drains the pending mirrors into the tracer
    assert np.array_equal(tr.get_eam(seq), eam[0].astype(np.float64)), "the engine-fed EAM is the routing that happened"
    eng.set_predictor(None)
    tr.finish_entry(seq)
    eng.close()
