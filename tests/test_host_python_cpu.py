"""Host-side Python mirrors that need no GPU: config objects, tracer/predictor/prefetcher wrappers
driving a recording stand-in for the engine."""
import json

import numpy as np
import pytest
import torch

from moe_infinity_amd import config as Cf
from moe_infinity_amd.memory import ExpertPredictor, ExpertPrefetcher, ExpertTracer


def test_archer_config_keys_and_defaults(tmp_path):
    c = Cf.ArcherConfig.load_from_json({"offload_path": str(tmp_path), "device_memory_ratio": 0.75, "unknown_key": 1})
    assert c.device_memory_ratio == 0.75 and c.trace_capacity == 1000 and c.num_threads == 8 and c.prefetch is False
    assert c.perfect_cache_file.endswith("perfect_cache")
    p = tmp_path / "cfg.json"
    p.write_text(json.dumps({"offload_path": "x", "device_memory_bytes": 123, "cache_policy": "lru"}))
    c = Cf.ArcherConfig.load_from_file(str(p))
    assert c.device_memory_bytes == 123 and c.cache_policy == "lru" and c.device_memory_ratio == 0.9


def test_model_presets_match_survey_section_8():
    m = Cf.mixtral_8x7b()
    assert (m.num_layers, m.num_experts, m.top_k, m.hidden, m.inter) == (32, 8, 2, 4096, 14336)
    d = Cf.deepseek_v2_lite()
    assert (d.num_layers, d.num_experts, d.top_k, d.hidden, d.inter, d.shared_inter) == (26, 64, 6, 2048, 1408, 2816)
    s = Cf.switch_base_8()
    assert (s.num_layers, s.num_experts, s.top_k, s.dtype, s.expert_capacity) == (12, 8, 1, Cf.DTYPE_F32, 64)
    n = Cf.nllb_moe_54b()
    assert (n.num_layers, n.num_experts, n.top_k, n.hidden, n.inter) == (12, 128, 2, 2048, 8192)
    # expert_type ids are the reference's (core/parallel/expert_module.h:13-18)
    assert (Cf.EXPERT_SWITCH, Cf.EXPERT_NLLB, Cf.EXPERT_MIXTRAL, Cf.EXPERT_DEEPSEEK) == (0, 2, 4, 5)


class _RecordingEngine:
    def __init__(self):
        self.calls = []

    def protect(self, pairs):
        self.calls.append(("protect", list(pairs)))

    def prefetch(self, layer, experts, scores=None):
        self.calls.append(("prefetch", layer, list(experts)))


def test_prefetcher_issues_the_reference_order_and_filters():
    L, E = 4, 6
    tr = ExpertTracer(4, L, E)
    hist = np.zeros((4, L, E), np.float32)
    hist[:, :, 0] = 10
    hist[:, :, 1] = 5
    hist[:, :, 2] = 1
    tr.load_trace(hist)
    pred = ExpertPredictor(L, E)
    pred.add_tracer(tr)
    pf = ExpertPrefetcher(L, E, tr)
    eng = _RecordingEngine()
    pf.set_archer_engine(eng)
    seq = tr.create_entry()
    m = pred.predict(seq, [0, 1], 0)
    assert m.shape == (L, E) and (m[0] > 0).all() and m[1, 0] > m[1, 1] > m[1, 2]
    ls, es, sc = pf.prefetch_experts(1, m)
    # replace_cache_candidates(all) first, then enqueue_prefetch in descending score; layer decay puts layer 1 first
    assert eng.calls[0][0] == "protect" and len(eng.calls[0][1]) == len(ls) == 3 * E
    assert (np.diff(sc) <= 0).all() and ls[0] == 1 and es[0] == 0
    flat = [(c[1], e) for c in eng.calls[1:] for e in c[2]]
    assert flat == list(zip(ls.tolist(), es.tolist()))
    eng.calls.clear()
    ls2, es2, _ = pf.prefetch_experts(1, m, max_experts=4, min_share=0.2, lookahead=2)
    assert len(ls2) == 4 and set(ls2.tolist()) <= {1, 2} and set(es2.tolist()) <= {0, 1}
    tr.finish_entry(seq)


class _RecordingEngine:
    """Stands in for MoEEngine on a GPU-less host: records what the adapter asks for."""

    class cfg:
        device_id = 0
        num_layers = 3
        num_experts = 4

    def __init__(self):
        self.calls = []
        self.resident = set()

    def protect(self, pairs):
        self.calls.append(("protect", [tuple(p) for p in pairs]))

    def prefetch(self, layer, experts, scores=None):
        self.calls.append(("prefetch", layer, list(experts)))
        self.resident.update((layer, e) for e in experts)

    def is_resident(self, layer, expert):
        return (layer, expert) in self.resident

    def expert_counters(self):
        c = np.zeros((3, 4, 6), np.int64)
        c[1, 2] = [5, 3, 2, 1, 3, 1]
        return c

    def clear_expert_cache_counts(self):
        self.calls.append(("clear",))

    def close(self):
        self.calls.append(("close",))


def test_prefetch_handle_adapter_keeps_the_reference_method_names_and_order():
    from moe_infinity_amd.prefetch_handle import PrefetchHandle

    eng = _RecordingEngine()
    tmap = {(l, e): 100 + 10 * l + e for l in range(3) for e in range(4)}  # expert_tensor_map of the reference
    h = PrefetchHandle(eng, tmap)
    h.register_expert_tensors(1, 2, [912, 913])  # the expert's other parameter ids
    h.replace_cache_candidates([tmap[(2, 1)], tmap[(1, 2)], 913])
    assert eng.calls[-1] == ("protect", [(2, 1), (1, 2)]), "ids of one expert collapse, order kept"
    assert h.get_node_default_device([tmap[(2, 1)]]) == 0
    assert h.get_node_device([tmap[(2, 1)]]) == -1 and not h.is_tensor_on_device(tmap[(2, 1)])
    h.enqueue_prefetch(tmap[(2, 1)], 0)
    assert eng.calls[-1] == ("prefetch", 2, [1]) and h.is_tensor_on_device(tmap[(2, 1)]) and h.get_node_device([tmap[(2, 1)]]) == 0
    hr = h.get_hit_rate()
    assert hr.shape == (12, 11) and hr.dtype == torch.int64
    row = hr[1 * 4 + 2]  # columns of model_topology.cpp:253-263
    assert [int(v) for v in row] == [5, 5, 0, 3, 3, 0, 3, 1, 0, 0, 1]
    with pytest.raises(KeyError):
        h.enqueue_prefetch(7, 0)
    with pytest.raises(ValueError):
        h.enqueue_prefetch(tmap[(0, 0)], 3)
    assert not hasattr(h, "begin")  # the loader / dense-layer methods live on moe_infinity_amd.prefetch_op.prefetch_handle
    assert h.prefetch_tensors(0, [1]) is None
    h.clean_up_resources()
    assert eng.calls[-1] == ("close",)


def test_reference_expert_prefetcher_drives_the_adapter_unmodified():
    """The reference's own memory/expert_prefetcher.py (imported from /root/reference when it is there; CPU-only
    build container) against the adapter: same protected set and the same prefetch order as our mirror class."""
    import importlib.util
    import os
    import sys
    import types

    src = "/root/reference/moe_infinity/memory/expert_prefetcher.py"
    if not os.path.exists(src):
        pytest.skip("reference checkout not present")
    from moe_infinity_amd.memory import ExpertPrefetcher, ExpertTracer
    from moe_infinity_amd.prefetch_handle import PrefetchHandle

    # the module imports moe_infinity.utils.parse_moe_param; stub the package around the one file we want
    L, E = 3, 4
    stub = types.ModuleType("moe_infinity")
    utils = types.ModuleType("moe_infinity.utils")
    utils.parse_moe_param = lambda config: (L, E, 0)
    stub.utils = utils
    saved = {k: sys.modules.get(k) for k in ("moe_infinity", "moe_infinity.utils")}
    sys.modules["moe_infinity"], sys.modules["moe_infinity.utils"] = stub, utils
    sys.dont_write_bytecode = True
    try:
        spec = importlib.util.spec_from_file_location("ref_expert_prefetcher", src)
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
    finally:
        for k, v in saved.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v
    tmap = {(l, e): 100 + 10 * l + e for l in range(L) for e in range(E)}
    m = np.array([[0, 0, 0, 0], [0.5, 0, 0.25, 0], [0, 0.125, 0, 0.75]])
    ref_eng = _RecordingEngine()
    ref = mod.ExpertPrefetcher(config=None)
    ref.set_archer_engine(PrefetchHandle(ref_eng, tmap))
    ref.expert_tensor_map = tmap
    ref.prefetch_experts(1, m)
    ours_eng = _RecordingEngine()
    ours = ExpertPrefetcher(L, E, ExpertTracer(4, L, E))
    ours.set_archer_engine(ours_eng)
    ours.prefetch_experts(1, m)

    def flat(calls):
        prot = [c[1] for c in calls if c[0] == "protect"]
        pre = [(c[1], e) for c in calls if c[0] == "prefetch" for e in c[2]]
        return prot, pre

    assert flat(ref_eng.calls) == flat(ours_eng.calls)
    assert flat(ref_eng.calls)[1] == [(2, 3), (1, 0), (1, 2), (2, 1)], "descending predicted share"


def test_the_deepseek_v3_gate_is_its_own_router_kind_and_other_mixes_are_refused():
    """modeling_deepseek_v3/modeling_deepseek.py:466-528 scores with a sigmoid plus a correction bias and picks groups by their
    top-2 sum: MOEINF_ROUTER_DEEPSEEK_V3 since round 6.  Any other mix of scoring function and top-k method (a sigmoid with V2's
    greedy rule, noaux_tc over softmax scores) must still fail loudly instead of being routed with V2's softmax rule."""
    import types

    from moe_infinity_amd.blocks import DeepseekMoEBlock
    base = dict(hidden_size=64, moe_intermediate_size=32, n_routed_experts=8, num_experts_per_tok=2, n_shared_experts=1,
                norm_topk_prob=False, routed_scaling_factor=1.0, n_group=None, topk_group=None)
    ok = DeepseekMoEBlock.engine_config(types.SimpleNamespace(topk_method="greedy", **base), 1, max_tokens=4)
    assert ok.router_kind == Cf.ROUTER_DEEPSEEK and ok.num_experts == 8
    v3 = DeepseekMoEBlock.engine_config(types.SimpleNamespace(**{**base, "topk_method": "noaux_tc", "scoring_func": "sigmoid", "n_group": 4, "topk_group": 2,
                                                                  "norm_topk_prob": True, "routed_scaling_factor": 2.5}), 1, max_tokens=4)
    assert v3.router_kind == Cf.ROUTER_DEEPSEEK_V3 and (v3.n_group, v3.topk_group, v3.norm_topk_prob, v3.routed_scaling_factor) == (4, 2, True, 2.5)
    for bad in (dict(topk_method="greedy", scoring_func="sigmoid"), dict(topk_method="noaux_tc")):
        with pytest.raises(NotImplementedError, match="expert_dispatcher"):
            DeepseekMoEBlock.engine_config(types.SimpleNamespace(**{**base, **bad}), 1, max_tokens=4)


def test_enqueue_expert_maps_a_cuda_ordinal_to_the_engine_that_serves_it():
    """dispatch_local passes gpu_id = expert_id % torch.cuda.device_count() (expert_executor.py:49-54).  A handle that drives
    fewer devices than the process can see must serve every such call from the expert's home engine, and devices=[2, 3]
    must look gpu_id up as an ordinal, not use it as an index (prefetch_op.py:507)."""
    from moe_infinity_amd.prefetch_op import slot_for_gpu

    # one process per GPU on an 8-GPU box: the handle drives device 3 only, dispatch_local names e % 8
    for e in range(16):
        assert slot_for_gpu([3], e % 8, 0) == 0
    # one process, two of four GPUs: ordinals 2 and 3 are engines 0 and 1; 0 and 1 are not driven -> home
    assert slot_for_gpu([2, 3], 3, 0) == 1 and slot_for_gpu([2, 3], 2, 1) == 0
    assert slot_for_gpu([2, 3], 0, 1) == 1 and slot_for_gpu([2, 3], 1, 0) == 0
    # the reference's own form: every GPU driven, gpu_id == engine index == ordinal
    for g in range(4):
        assert slot_for_gpu([0, 1, 2, 3], g, (g + 1) % 4) == g
    # several engines on one GPU (tests): gpu_id is the engine index; out of range -> home
    assert slot_for_gpu([0, 0], 1, 0) == 1 and slot_for_gpu([0, 0], 0, 1) == 0 and slot_for_gpu([0, 0], 5, 1) == 1
    # an expert that was never dealt a home (cannot happen after register_expert) still lands on a valid engine
    assert slot_for_gpu([0, 1], 7, None) == 1 and slot_for_gpu([4], -1, None) == 0
