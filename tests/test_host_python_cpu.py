"""Host-side Python mirrors that need no GPU: config objects, tracer/predictor/prefetcher wrappers
driving a recording stand-in for the engine."""
import json

import numpy as np

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
