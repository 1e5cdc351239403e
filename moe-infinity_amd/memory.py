"""Mirrors of moe_infinity/memory/{expert_tracer,expert_predictor,expert_prefetcher}.py on the
native tracer (csrc/tracer.h) and the engine's prefetch/protect entry points.  Same class and method
names; EAMs are numpy arrays as in the reference.  No GPU work: the reference's cosine search on
``cuda:0`` (expert_tracer.py:94-125) is replaced by the host-side incremental search."""
from typing import Dict, Tuple

import numpy as np

from .engine import ExpertTracerNative


class ExpertTracer:
    def __init__(self, capacity: int, num_layers: int, num_experts: int, num_encoder_layers: int = 0):
        self.num_layers, self.num_experts, self.num_encoder_layers = num_layers, num_experts, num_encoder_layers
        self.capacity = capacity
        self._native = ExpertTracerNative(num_layers, num_experts, capacity)

    def load_trace(self, trace):
        if not isinstance(trace, np.ndarray):
            trace = np.load(trace, allow_pickle=False)
        assert trace.shape[0] <= self.capacity, "loaded trace capacity must be <= capacity in config"
        self._native.load_trace(trace)

    def create_entry(self):
        return self._native.create_entry()

    def finish_entry(self, seq_id):
        self._native.finish_entry(seq_id)

    def get_entry_matrix(self, seq_id) -> np.ndarray:
        return self._native.get_eam(seq_id)


class ExpertPredictor:
    def __init__(self, num_layers: int, num_experts: int):
        self.num_layers, self.num_experts = num_layers, num_experts

    def add_tracer(self, tracer: ExpertTracer):
        self.tracer = tracer

    def predict(self, seq_id, expert_list, layer_idx) -> np.ndarray:
        """update_entry + find_most_similar + layer decay (expert_predictor.py:17-35)."""
        m, _ = self.tracer._native.predict(seq_id, np.asarray(expert_list).reshape(-1), layer_idx)
        return m


class ExpertPrefetcher:
    def __init__(self, num_layers: int, num_experts: int, tracer: ExpertTracer):
        self.num_layers, self.num_experts = num_layers, num_experts
        self._native = tracer._native
        self.archer_engine = None

    def set_archer_engine(self, engine):
        self.archer_engine = engine

    def prefetch_experts_list(self, layer_id, expert_list):
        self.archer_engine.prefetch(layer_id, list(expert_list))

    def fetch_experts_lock_cache(self, layer_id, expert_list):
        self.archer_engine.protect([(layer_id, int(e)) for e in expert_list])

    def prefetch_experts(self, layer_id, expert_matrix, max_experts=None, min_share=0.0, lookahead=None):
        """expert_prefetcher.py:42-59: every (layer >= layer_id, expert) with a positive score, by
        descending score: replace_cache_candidates(all) then enqueue_prefetch(each) in that order."""
        ls, es, sc = self._native.prefetch_order(layer_id, expert_matrix)
        # additions (the reference enqueues everything it predicts, which thrashes a small cache):
        #   min_share  keep an expert only if it holds >= this share of its layer's predicted activations
        #   lookahead  keep only layers < layer_id + lookahead
        #   max_experts cap the list
        if min_share > 0.0 or lookahead is not None:
            m = np.asarray(expert_matrix, dtype=np.float64)
            share = m / np.maximum(m.sum(axis=1, keepdims=True), 1e-30)
            keep = np.array([share[l, e] >= min_share and (lookahead is None or l < layer_id + lookahead)
                             for l, e in zip(ls, es)], dtype=bool) if len(ls) else np.zeros(0, bool)
            ls, es, sc = ls[keep], es[keep], sc[keep]
        if max_experts is not None:
            ls, es, sc = ls[:max_experts], es[:max_experts], sc[:max_experts]
        self.archer_engine.protect(list(zip(ls.tolist(), es.tolist())))
        i = 0
        while i < len(ls):  # the C ABI takes one layer per call; keep the global priority order
            j = i
            while j < len(ls) and ls[j] == ls[i]:
                j += 1
            self.archer_engine.prefetch(int(ls[i]), es[i:j].tolist(), sc[i:j].tolist())
            i = j
        return ls, es, sc

    def prefetch_experts_by_priority(self, layer_id, expert_freq, eam, max_experts=None):
        """Speculative requests for the layers AFTER ``layer_id`` ordered by the reference's ``priority_score``
        (expert_priority_score.py:85-172: layer decay x the running sequence's EAM x visit frequency) — the scoring
        module the reference ships but never calls.  Scores travel to the engine with the request, where they pick
        the queue level (csrc/prefetch_queue.h), so a later, better-scored request overtakes a waiting one.
        ``expert_freq``: {(expert, layer): visits} (e.g. from get_hit_rate); ``eam``: the running sequence's [L, E]
        activation matrix (ExpertTracer.get_entry_matrix)."""
        from . import priority_score as PS

        entry = PS.ExpertTraceEntry("running", np.array(eam, dtype=np.float64, copy=True), 1, 0)
        scored = [c for c in PS.priority_score(expert_freq, set(), set(), entry, layer_id, self.num_layers, inplace=False)
                  if c.layer_idx > layer_id]
        scored.sort(key=lambda c: (-c.r, c.layer_idx, c.expert_idx))
        if max_experts is not None:
            scored = scored[:max_experts]
        if not scored:
            return []
        top = scored[0].r
        self.archer_engine.protect([(c.layer_idx, c.expert_idx) for c in scored])
        i = 0
        while i < len(scored):  # one layer per C-ABI call, global order kept
            j = i
            while j < len(scored) and scored[j].layer_idx == scored[i].layer_idx:
                j += 1
            self.archer_engine.prefetch(scored[i].layer_idx, [c.expert_idx for c in scored[i:j]], [float(c.r / top) for c in scored[i:j]])
            i = j
        return scored
