"""The drop-in ``prefetch_op`` module (moe-infinity_amd/prefetch_op.py) driven the way the reference's OffloadEngine
drives the pybind module (moe_infinity/runtime/model_offload.py:143-145,471-477,751-873,883-973): offload a state
dict, register placeholders, set_topology, expert_dispatcher(E, L, dtype, type, threads).register_expert(layer,
expert, tensor_ids), begin/end around dense modules, dispatch_local for the experts.  -m gpu."""
import pytest
import torch
import torch.nn.functional as F

from helpers import R, acts, assert_block_close, make_weights

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


class _Param:
    """a parameter of an empty-initialised model: 1-element pinned-style placeholder (apply_to_model_decorator,
    model_offload.py:183-205)"""

    def __init__(self, dtype):
        self.p = torch.nn.Parameter(torch.zeros(1, dtype=dtype), requires_grad=False)


def _build(tmp_path, L, E, H, Fd, seed, dense_cache_fraction=0.7, **opts):
    from moe_infinity_amd import prefetch_op as P

    P.configure(dense_cache_fraction=dense_cache_fraction, device_memory_bytes=opts.get("device_memory_bytes", 0), max_tokens=8)
    handle = P.prefetch_handle(str(tmp_path), 0.5)
    dt = torch.bfloat16
    g = torch.Generator().manual_seed(seed)
    state, topo, model = {}, [], {}
    next_id = [0]

    def add(name, tensor):
        state[name] = (next_id[0], tensor)
        next_id[0] += 1
        return state[name][0]

    layers = []
    for l in range(L):
        attn = (torch.randn(H, H, generator=g) * 0.05).to(dt)
        bias = (torch.randn(H, generator=g) * 0.05).to(dt)
        gate, experts, _ = make_weights("mixtral", H, Fd, E, seed + 10 * l, dt)
        ids_attn = [add(f"layers.{l}.attn.weight", attn), add(f"layers.{l}.attn.bias", bias)]
        id_gate = add(f"layers.{l}.moe.gate.weight", gate)
        ex_ids = [[add(f"layers.{l}.moe.experts.{e}.w1.weight", experts[e][0]), add(f"layers.{l}.moe.experts.{e}.w2.weight", experts[e][1]),
                   add(f"layers.{l}.moe.experts.{e}.w3.weight", experts[e][2])] for e in range(E)]
        topo += [(f"layers.{l}.attn", [ids_attn]), (f"layers.{l}.moe.gate", [[id_gate]]), (f"layers.{l}.moe.experts", ex_ids)]
        layers.append(dict(attn=attn, bias=bias, gate=gate, experts=experts, ids_attn=ids_attn, id_gate=id_gate, ex_ids=ex_ids))
    assert not handle.is_tensor_index_initialized()
    for name, (tid, t) in state.items():  # _offload_state_dict (model_offload.py:891-906)
        if not handle.is_tensor_offloaded(tid):
            handle.offload(t, tid)
    params = {}
    for name, (tid, t) in state.items():  # setup_archer_hooks (model_offload.py:751-766)
        params[tid] = _Param(t.dtype).p
        handle.register(params[tid].data, tid)
    handle.set_topology(topo)
    disp = P.expert_dispatcher(E, L, 0, 4, 8)  # (num_experts, num_layers, dtype bf16, MIXTRAL, num_threads)
    for l, lay in enumerate(layers):
        for e in range(E):
            assert handle.get_node_default_device(lay["ex_ids"][e]) == 0
            disp.register_expert(l, e, lay["ex_ids"][e])
    return P, handle, disp, layers, params


def _forward(handle, disp, layers, params, x, k=2):
    from moe_infinity_amd.expert_executor import DistributedExpertExecutor

    ex = DistributedExpertExecutor(None)
    ex.set_expert_dispatcher(disp)
    h = x.to(DEV)
    ref_h = x.clone()
    for l, lay in enumerate(layers):
        w, b = params[lay["ids_attn"][0]], params[lay["ids_attn"][1]]
        handle.begin(0, w)  # _pre_forward_module_hook (model_offload.py:925-947)
        handle.begin(0, b)
        assert w.is_cuda and tuple(w.shape) == tuple(lay["attn"].shape) and torch.equal(w.cpu(), lay["attn"]) and torch.equal(b.cpu(), lay["bias"])
        h = F.linear(h, w, b)
        handle.end(0, w)  # _post_forward_module_hook
        handle.end(0, b)
        assert w.numel() == 1 and not w.is_cuda
        ref_h = F.linear(ref_h, lay["attn"], lay["bias"])
        h = ref_h.to(DEV)  # same input for both sides from here (GPU vs CPU GEMM order is not what this test is about)
        gw = params[lay["id_gate"]]
        handle.begin(0, gw)
        assert torch.equal(gw.cpu(), lay["gate"])
        sel, wts, _ = R.route_mixtral(ref_h, gw.cpu(), k)  # the block's router (oracle arithmetic: deterministic)
        handle.end(0, gw)
        router_mask, weights_mask = R.masks_from_topk(sel, wts, len(lay["experts"]))
        res = ex.dispatch_local(h, router_mask.to(DEV), l)  # mixtral.py:87-94
        final = torch.zeros_like(ref_h)
        for out, layer, idx, _hit in res:  # mixtral.py:96-101
            tok = router_mask[:, idx]
            final[tok, :] += out.cpu() * weights_mask[tok, idx][:, None]
        ref = R.block_mixtral(ref_h[None], lay["gate"], lay["experts"], top_k=k)
        assert_block_close(final, ref, torch.bfloat16, f"layer {l} MoE block through prefetch_op")
        ref_h = ref.out[0]
        h = ref_h.to(DEV)
    return ref_h


def test_offload_engine_flow_through_prefetch_op(tmp_path):
    L, E, H, Fd = 3, 4, 256, 512
    P, handle, disp, layers, params = _build(tmp_path, L, E, H, Fd, 2100)
    try:
        x = acts(5, H, torch.bfloat16, 2101)
        _forward(handle, disp, layers, params, x)
        _forward(handle, disp, layers, params, x)  # second pass: experts resident
        hr = handle.get_hit_rate()
        assert tuple(hr.shape) == (L * (2 + E), 11) and int(hr[:, 10].sum()) == L * E
        assert handle.get_node_device(layers[0]["ids_attn"]) == 0 and handle.is_tensor_on_device(layers[0]["id_gate"])
        tr = torch.arange((L - 1) * E * E, dtype=torch.int64).reshape(L - 1, E, E)
        handle.set_trace(tr)
        assert torch.equal(handle.get_trace(), tr)
        with pytest.raises(ValueError):
            handle.set_trace(torch.zeros(2, 2, dtype=torch.int64))
        # the prefetcher's calls (memory/expert_prefetcher.py:28-59) by tensor id
        ids = [layers[1]["ex_ids"][e][0] for e in range(E)]
        handle.replace_cache_candidates(ids)
        for tid in ids:
            handle.enqueue_prefetch(tid, handle.get_node_default_device([tid]))
        disp.clear_expert_cache_counts()
        with pytest.raises(RuntimeError):
            disp.register_expert(0, 0, [layers[0]["ex_ids"][0][0], layers[0]["ex_ids"][1][0]])  # ids of two nodes
    finally:
        handle.clean_up_resources()
    # a second process start on the same directory finds the index (the "Loading model from offload_path" branch)
    h2 = P.prefetch_handle(str(tmp_path), 0.5)
    try:
        assert h2.is_tensor_index_initialized() and h2.is_tensor_offloaded(0) and not h2.is_tensor_offloaded(10 ** 6)
    finally:
        h2.clean_up_resources()


def test_dense_nodes_over_the_cache_limit_are_dropped_and_refetched(tmp_path):
    """RemoveCachedDenseNode (task_scheduler.cpp:319-378): with a dense cache limit that holds only a few nodes the
    earliest layers are dropped; begin() brings them back from the offload directory; results do not change."""
    L, E, H, Fd = 4, 4, 256, 512
    total = torch.cuda.mem_get_info(0)[1]
    frac = (2.5 * (H * H * 2 + 4096)) / total  # room for ~2 attention nodes
    P, handle, disp, layers, params = _build(tmp_path, L, E, H, Fd, 2200, dense_cache_fraction=frac, device_memory_bytes=6 * 3 * Fd * H * 2)
    try:
        on_dev = sum(handle.get_node_device(lay["ids_attn"]) >= 0 for lay in layers)
        assert on_dev < L, "the limit must have forced some dense nodes out"
        x = acts(4, H, torch.bfloat16, 2201)
        for _ in range(2):
            _forward(handle, disp, layers, params, x)
        st = handle.engine.stats()
        assert st["expert_misses"] > 0 and st["evictions"] > 0  # the 6-slot expert cache was exercised too
    finally:
        handle.clean_up_resources()
        P.configure(dense_cache_fraction=0.7, device_memory_bytes=0)
