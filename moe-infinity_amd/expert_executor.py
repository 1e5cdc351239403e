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
    def __init__(self, engine):
        self.engine = engine
        self._queue = []
        self._expected = 0
        self._hidden = None
        self._mask = None

    # register_expert(layer, expert, tensor_ids) in the reference; here the tensors themselves
    def register_expert(self, layer_idx: int, expert_idx: int, tensors: Sequence[torch.Tensor]):
        self.engine.register_expert(layer_idx, expert_idx, tensors)

    def set_inputs(self, hidden_states: torch.Tensor, router_mask: torch.Tensor):
        self._hidden = hidden_states.reshape(-1, hidden_states.shape[-1]).contiguous()
        self._mask = router_mask.reshape(-1, router_mask.shape[-1])

    def set_expected_queue(self, expected_pending: int):
        self._expected = expected_pending

    def enqueue_expert(self, layer_idx: int, expert_idx: int, gpu_id: int = 0, remote: bool = False):
        self._queue.append((layer_idx, expert_idx))

    def wait_expert(self) -> List[Tuple[torch.Tensor, int, int, int]]:
        if len(self._queue) != self._expected:
            raise RuntimeError(f"expected {self._expected} enqueued experts, got {len(self._queue)}")
        if not self._queue:
            return []
        layers = {l for l, _ in self._queue}
        if len(layers) != 1:
            raise RuntimeError("one wait_expert() serves one layer (as dispatch_local uses it)")
        layer = layers.pop()
        enq = sorted({e for _, e in self._queue})
        mask = self._mask
        if len(enq) != mask.shape[1]:  # only enqueued experts run
            keep = torch.zeros(mask.shape[1], dtype=torch.bool, device=mask.device)
            keep[torch.tensor(enq, device=mask.device)] = True
            mask = mask.bool() & keep
        y, counts, hit = self.engine.dispatch_mask(layer, self._hidden, mask)
        out, row = [], 0
        for e in range(len(counts)):
            if counts[e] > 0:
                out.append((y[row:row + counts[e]], layer, e, int(hit[e])))
                row += int(counts[e])
        self._queue = []
        return out

    def clear_expert_cache_counts(self):
        self.engine.clear_expert_cache_counts()


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
