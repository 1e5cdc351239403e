"""``PrefetchHandle`` — the hot-path subset of the pybind class ``prefetch_op.prefetch_handle``
(core/python/py_archer_prefetch.cpp:14-80) under the reference's own method names, on top of ``MoEEngine``.

The reference addresses experts by *tensor id* (the first parameter id of the expert:
``expert_tensor_map[(layer, expert)]``, model_offload.py:866-871, used by memory/expert_prefetcher.py:28-59);
the engine addresses them by ``(layer, expert)``.  This adapter keeps the id -> (layer, expert) table, so the
reference's ``ExpertPrefetcher`` can drive the engine **unmodified**:

    prefetcher = moe_infinity.memory.ExpertPrefetcher(hf_config)
    prefetcher.set_archer_engine(PrefetchHandle(engine, expert_tensor_map))
    prefetcher.expert_tensor_map = expert_tensor_map

This adapter covers the expert-cache methods only (an engine that was built directly from tensors has no offload
directory or topology).  The FULL pybind surface — ``prefetch_handle(prefix, device_memory_ratio)`` with ``offload``,
``register``, ``begin``/``end``, ``set_topology``, ``update_tensor_map``, ``fetch_tensors`` ... and
``expert_dispatcher(num_experts, num_layers, dtype, expert_type, num_threads)`` — is ``moe_infinity_amd.prefetch_op``.
"""
from typing import Dict, Iterable, List, Sequence, Tuple

import numpy as np
import torch


class PrefetchHandle:
    def __init__(self, engine, expert_tensor_map: Dict[Tuple[int, int], int] = None):
        self.engine = engine
        self._by_id: Dict[int, Tuple[int, int]] = {}
        if expert_tensor_map:
            for (layer, expert), tid in expert_tensor_map.items():
                self.register_expert_tensors(layer, expert, [tid])

    # -- id table (the reference builds it in OffloadEngine.get_topology, model_offload.py:851-871)
    def register_expert_tensors(self, layer: int, expert: int, tensor_ids: Iterable[int]):
        for tid in tensor_ids:
            self._by_id[int(tid)] = (int(layer), int(expert))

    def _pair(self, tensor_id) -> Tuple[int, int]:
        try:
            return self._by_id[int(tensor_id)]
        except KeyError:
            raise KeyError(f"tensor id {tensor_id} does not belong to a registered expert") from None

    def _pairs(self, tensor_ids: Sequence[int]) -> List[Tuple[int, int]]:
        seen, out = set(), []
        for tid in tensor_ids:  # several ids of one expert collapse to one entry, first position kept
            p = self._pair(tid)
            if p not in seen:
                seen.add(p)
                out.append(p)
        return out

    # -- prefetch_handle methods on this path
    def replace_cache_candidates(self, tensor_ids: Sequence[int]):
        """archer_prefetch_handle.cpp:195-205 -> ArcherTaskPool::ReplaceCacheCandidates: the protected set."""
        self.engine.protect(self._pairs(tensor_ids))

    def enqueue_prefetch(self, tensor_id: int, gpu_id: int = 0):
        """archer_prefetch_handle.cpp:207-218 -> EnqueueTask(priority 1).  gpu_id is the node's default device; the
        engine serves exactly one device, so it is checked, not used."""
        if gpu_id not in (-1, self.engine.cfg.device_id):
            raise ValueError(f"engine serves device {self.engine.cfg.device_id}, prefetch asked for {gpu_id}")
        layer, expert = self._pair(tensor_id)
        self.engine.prefetch(layer, [expert])

    def get_node_default_device(self, tensor_ids: Sequence[int]) -> int:
        self._pairs(tensor_ids)
        return int(self.engine.cfg.device_id)

    def get_node_device(self, tensor_ids: Sequence[int]) -> int:
        """Device index if the expert is resident, -1 (CPU) otherwise (py_archer_prefetch.cpp:62-68)."""
        pairs = self._pairs(tensor_ids)
        return int(self.engine.cfg.device_id) if all(self.engine.is_resident(l, e) for l, e in pairs) else -1

    def is_tensor_on_device(self, tensor_id) -> bool:
        layer, expert = self._pair(int(tensor_id))
        return self.engine.is_resident(layer, expert)

    def get_hit_rate(self) -> torch.Tensor:
        """int64 [N, 11] like ArcherPrefetchHandle::GetHitRate (archer_prefetch_handle.cpp:281-297); columns as
        ArcherTopologyHandle::GetNodeVisitCounts builds them (model_topology.cpp:253-263): visit, gpu_visit,
        cpu_visit, hit, gpu_hit, cpu_hit, #tensor ids, prefetch, unused_count, io_state, is_sparse.  One row per
        expert in (layer, expert) order (the reference also lists the dense nodes, which this engine does not
        manage).  There is no CPU execution here, so the cpu_* columns are 0."""
        c = self.engine.expert_counters()  # [L, E, 7] = visit, hit, miss, prefetch, incache, resident, unused
        L, E, _ = c.shape
        out = np.zeros((L * E, 11), np.int64)
        v, h, _m, p = (c[..., i].reshape(-1) for i in range(4))
        out[:, 0] = v
        out[:, 1] = v
        out[:, 3] = h
        out[:, 4] = h
        nids = np.zeros((L, E), np.int64)
        for (layer, expert) in self._by_id.values():
            if layer < L and expert < E:
                nids[layer, expert] += 1
        out[:, 6] = nids.reshape(-1)
        out[:, 7] = p
        out[:, 10] = 1
        return torch.from_numpy(out)

    def clear_expert_cache_counts(self):
        self.engine.clear_expert_cache_counts()

    def clean_up_resources(self):
        self.engine.close()

    def prefetch_tensors(self, request_id, tensor_ids):
        """A no-op in the reference too (archer_prefetch_handle.cpp:182-193)."""
        return None
