"""Expert priority scores — host-side mirror of ``moe_infinity/memory/expert_priority_score.py`` and
``moe_infinity/memory/expert_entry.py`` ("unused priority scoring", reference lines 84-172).

Same names, arguments and results as the reference module, so code written against it runs unchanged; the bodies are
array formulas instead of per-layer Python loops.  What each function returns (reference line numbers):

  lru_score(cache_entries)                        :24-30   r = entry.timestamp
  lru_score_with_layers(cache_entries, layer)     :33-50   r = 1e10 for the next three layers, else timestamp
  lfu_score(expert_freq)                          :53-66   r = count / total (total 0 -> 1)
  oracle_score(expert_freq, decoder_entry)        :69-82   r = count / total + 1e-6 (no counts at all: 1 + 1e-6)
  priority_score(freq, cache, traces, dec, l, L)  :85-172  r = topo[l'] * decoder[l', e] * frequency[l', e]
      frequency  counts normalised over the whole matrix (+1e-6); an all-zero encoder or decoder half counts as ones
      topo       layer decay: layers already executed in the running half 1.0, later layers of that half
                 1 - i/Le, the other half (i - Le)/(Le + 1) resp. 1 - i/Le; normalised over the matrix (+1e-6)
      decoder    the running sequence's EAM, every row normalised to sum 1 (all-zero rows: uniform), then the matrix
                 normalised to sum 1 (+1e-6)
  The reference also accumulates a ``seq_expert_score`` from the trace entries and then leaves it out of the product;
  so does this module (the argument is accepted and ignored).  Like the reference, ``priority_score`` row-normalises
  ``decoder_entry.matrix`` IN PLACE unless ``inplace=False``.

``levels_from_scores`` is the bridge to the engine: it turns such scores into the 1..19 speculative levels of the
pending-transfer queue (csrc/prefetch_queue.h) the way ``moeinf_prefetch`` maps a score in (0, 1].
"""
from dataclasses import dataclass
from typing import Dict, Iterable, List, Tuple

import numpy as np


@dataclass
class ExpertTraceEntry:  # expert_entry.py:6-14
    seq_id: str = None
    matrix: np.ndarray = None
    access: int = 0
    num_new_tokens: int = 0

    def __hash__(self):
        return hash(self.seq_id)


@dataclass
class ExpertCacheEntry:  # expert_entry.py:17-26
    expert_idx: int = None
    layer_idx: int = None
    r: float = 0.0
    visit: int = 0
    timestamp: int = 0

    def __hash__(self):
        return hash((self.layer_idx, self.expert_idx))


def convert_score_matrix_to_list(score_matrix: np.ndarray) -> List[ExpertCacheEntry]:
    """positive entries of a [L, E] score matrix, layer-major"""
    ls, es = np.nonzero(np.asarray(score_matrix) > 0)
    return [ExpertCacheEntry(int(e), int(l), score_matrix[l, e]) for l, e in zip(ls, es)]


def lru_score(cache_entries: Iterable[ExpertCacheEntry]) -> List[ExpertCacheEntry]:
    return [ExpertCacheEntry(c.expert_idx, c.layer_idx, c.timestamp) for c in cache_entries]


def lru_score_with_layers(cache_entries: Iterable[ExpertCacheEntry], current_layer: int) -> List[ExpertCacheEntry]:
    near = lambda c: current_layer <= c.layer_idx < current_layer + 3  # noqa: E731
    return [ExpertCacheEntry(c.expert_idx, c.layer_idx, 1e10 if near(c) else c.timestamp) for c in cache_entries]


def lfu_score(expert_freq: Dict[Tuple[int, int], float]) -> List[ExpertCacheEntry]:
    total = 0
    for v in expert_freq.values():
        total += v
    if total == 0:
        total = 1
    return [ExpertCacheEntry(e, l, v / total) for (e, l), v in expert_freq.items()]


def _frequency_matrix(expert_freq, shape, dtype):
    f = np.zeros(shape, dtype=dtype)
    total = 0
    for (e, l), v in expert_freq.items():
        f[l, e] = v
        total += v
    return f, total


def oracle_score(expert_freq: Dict[Tuple[int, int], float], decoder_entry: ExpertTraceEntry) -> List[ExpertCacheEntry]:
    f, total = _frequency_matrix(expert_freq, decoder_entry.matrix.shape, decoder_entry.matrix.dtype)
    if total == 0:
        total = 1
        f = np.ones_like(f)
    return convert_score_matrix_to_list(f / total + 1e-6)


def topo_score(num_layers: int, num_experts: int, current_layer: int, total_layer: int, dtype=np.float64) -> np.ndarray:
    """the layer-decay factor of ``priority_score`` before normalisation, [L, E]"""
    le = total_layer // 2
    i = np.arange(num_layers, dtype=np.float64)
    first = -1.0 / le * i + 1.0            # decay_from_first(i, le)
    last = 1.0 / (le + 1) * (i - le)       # decay_from_last(i - le, le)
    enc = i < le
    if current_layer < le:
        col = np.where(enc, np.where(i > current_layer, first, 1.0), last)
    else:
        col = np.where(enc, first, np.where(i > current_layer, last, 1.0))
    return np.repeat(col[:, None], num_experts, axis=1).astype(dtype)


def priority_score(expert_freq, cache_entries, trace_entries, decoder_entry: ExpertTraceEntry, current_layer: int,
                   total_layer: int, inplace: bool = True) -> List[ExpertCacheEntry]:
    m = decoder_entry.matrix
    le = total_layer // 2
    f, _ = _frequency_matrix(expert_freq, m.shape, m.dtype)
    if np.sum(f[le:]) == 0:
        f[le:] = 1
    if np.sum(f[:le]) == 0:
        f[:le] = 1
    f = f / np.sum(f) + 1e-6
    topo = topo_score(m.shape[0], m.shape[1], current_layer, total_layer, m.dtype)
    topo = topo / np.sum(topo) + 1e-6
    d = m if inplace else m.copy()
    if np.sum(d) == 0:
        d = np.ones_like(d)
    zero_rows = d.sum(axis=1) == 0
    d[zero_rows, :] = 1
    d /= d.sum(axis=1, keepdims=True)
    d = d / np.sum(d) + 1e-6
    return convert_score_matrix_to_list(topo * d * f)


def score_matrix(entries: Iterable[ExpertCacheEntry], num_layers: int, num_experts: int) -> np.ndarray:
    """[L, E] matrix of the r values of a score list (0 where absent)"""
    out = np.zeros((num_layers, num_experts), dtype=np.float64)
    for c in entries:
        out[c.layer_idx, c.expert_idx] = c.r
    return out


def levels_from_scores(scores: np.ndarray) -> np.ndarray:
    """queue level (1 = most urgent speculative level .. 19) of every score, relative to the largest score: the same map
    as ``moeinf_priority_from_score`` applied to score / max(score)"""
    s = np.asarray(scores, dtype=np.float64)
    top = float(s.max()) if s.size else 0.0
    rel = np.clip(s / top, 0.0, 1.0) if top > 0 else np.zeros_like(s)
    return np.clip(1 + np.floor((1.0 - rel) * 18.0 + 0.5), 1, 19).astype(np.int32)
