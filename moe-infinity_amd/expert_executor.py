"""Mirrors of the reference's dispatcher objects for callers that keep its PYTHON router and combine.

* ``ExpertDispatcher``  = the pybind class ``prefetch_op.expert_dispatcher``
  (core/python/py_archer_prefetch.cpp:84-92, core/parallel/expert_dispatcher.h:27-137): same method
  names, argument meaning and result tuples.  Instead of per-expert threads it issues ONE grouped
  launch per FFN stage for everything enqueued (moeinf_dispatch_mask).
* ``DistributedExpertExecutor.dispatch_local`` = moe_infinity/distributed/expert_executor.py:32-58.
"""
from typing import List, Sequence, Tuple

import numpy as np
import torch


class ExpertDispatcher:
    """``engine``: one MoEEngine, or a list of them — one per expert device, in ``gpu_id`` order (every engine holds the
    experts with ``expert % len(engines) == its index``, the reference's placement, model_topology.cpp:533-536).
    ``enqueue_expert``'s ``gpu_id`` picks the engine, modulo their number (expert_dispatcher.cpp:135-137 queues on
    ``gpu_id``); results come back on the hidden states' device (:405).  The drop-in over the reference's pybind
    constructor signature is ``prefetch_op.expert_dispatcher``."""

    def __init__(self, engine):
        self.engines = list(engine) if isinstance(engine, (list, tuple)) else [engine]
        self.engine = self.engines[0]
        self._queue = []
        self._expected = 0
        self._hidden = None
        self._mask = None

    # register_expert(layer, expert, tensor_ids) in the reference; here the tensors themselves
    def register_expert(self, layer_idx: int, expert_idx: int, tensors: Sequence[torch.Tensor]):
        self.engines[expert_idx % len(self.engines)].register_expert(layer_idx, expert_idx, tensors)

    def set_inputs(self, hidden_states: torch.Tensor, router_mask: torch.Tensor):
        self._hidden = hidden_states.reshape(-1, hidden_states.shape[-1]).contiguous()
        self._mask = router_mask.reshape(-1, router_mask.shape[-1])

    def set_expected_queue(self, expected_pending: int):
        self._expected = expected_pending

    def enqueue_expert(self, layer_idx: int, expert_idx: int, gpu_id: int = 0, remote: bool = False):
        if len(self.engines) > 1 and int(gpu_id) % len(self.engines) != expert_idx % len(self.engines):
            raise RuntimeError(f"expert {expert_idx} lives on engine {expert_idx % len(self.engines)}, not on gpu_id {gpu_id} "
                               "(this mirror does not move experts between devices; prefetch_op.expert_dispatcher does)")
        self._queue.append((layer_idx, expert_idx, expert_idx % len(self.engines)))

    def wait_expert(self) -> List[Tuple[torch.Tensor, int, int, int]]:
        if len(self._queue) != self._expected:
            raise RuntimeError(f"expected {self._expected} enqueued experts, got {len(self._queue)}")
        if not self._queue:
            return []
        layers = {l for l, _, _ in self._queue}
        if len(layers) != 1:
            raise RuntimeError("one wait_expert() serves one layer (as dispatch_local uses it)")
        layer = layers.pop()
        home = self._hidden.device
        out = []
        for slot in sorted({s for _, _, s in self._queue}):
            eng = self.engines[slot]
            enq = sorted({e for _, e, s in self._queue if s == slot})
            hidden = self._hidden if self._hidden.device == eng.device else self._hidden.to(eng.device)
            mask = self._mask if self._mask.device == eng.device else self._mask.to(eng.device)
            # only the experts enqueued on this engine run (moeinf_dispatch_mask_subset)
            y, counts, hit = eng.dispatch_mask(layer, hidden, mask, experts=None if len(enq) == mask.shape[1] else enq)
            if y.device != home:
                y = y.to(home)
            row = 0
            for e in range(len(counts)):
                if counts[e] > 0:
                    out.append((y[row:row + counts[e]], layer, e, int(hit[e])))
                    row += int(counts[e])
        out.sort(key=lambda r: r[2])
        self._queue = []
        return out

    def clear_expert_cache_counts(self):
        for eng in self.engines:
            eng.clear_expert_cache_counts()


class DistributedExpertExecutor:
    def __init__(self, archer_config=None):
        self.archer_config = archer_config

    def set_expert_dispatcher(self, expert_dispatcher: ExpertDispatcher):
        self.expert_dispatcher = expert_dispatcher

    def dispatch_local(self, hidden_states, router_mask, layer_id):
        num_expert = router_mask.shape[-1]
        expert_count = torch.sum(router_mask.view((-1, num_expert)), dim=0).cpu().numpy().flatten()
        expert_list = np.arange(num_expert).astype(int)[expert_count > 0].tolist()
        self.expert_dispatcher.set_inputs(hidden_states, router_mask)
        self.expert_dispatcher.set_expected_queue(len(expert_list))
        total_gpus = max(1, torch.cuda.device_count())
        for expert_id in expert_list:
            self.expert_dispatcher.enqueue_expert(layer_id, expert_id, expert_id % total_gpus, False)
        return self.expert_dispatcher.wait_expert()
