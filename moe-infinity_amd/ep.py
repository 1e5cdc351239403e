"""Expert parallelism across the GPUs of one node.

One process per GPU.  Expert e of every layer lives on rank (e % world_size) — the reference's
placement (core/model/model_topology.cpp:533-536, moe_infinity/distributed/expert_executor.py:49-54),
which the reference serves from ONE process with implicit P2P ``tensor.to(device)`` copies
(core/parallel/expert_dispatcher.cpp:284,405).  Here every rank routes its own tokens, and routed
rows travel with one all-to-all each way (RCCL over xGMI; torch.distributed backend "nccl"):

    route (local)  ->  pack rows by destination rank  ->  all_to_all  ->  grouped expert FFN on the
    owner  ->  all_to_all back  ->  deterministic combine (local)

Buffers have a fixed per-peer capacity (tokens*K rows: worst case every pair goes to one rank), so
the exchange needs no host-side size negotiation.  Every exchange row is H activations plus a
16-byte tail holding the row's expert id (-1 = padding), so payload and metadata cross the fabric
in ONE all-to-all per direction: at decode sizes the exchange is latency-bound (KBs per peer) and the
number of collectives is what matters.

``ops`` abstracts the four compute steps so the host logic here can be exercised on CPU with the
gloo backend (tests/test_ep_gloo.py supplies oracle-backed ops); the product ops are
``HipEpOps`` (HIP kernels through the C ABI).
"""
from typing import Optional

import torch
import torch.distributed as dist


class HipEpOps:
    """The four EP compute steps on the HIP engine (include/moeinf.h: moeinf_ep_*)."""

    def __init__(self, engine):
        self.engine = engine

    def route(self, layer, x2, gate_w):
        from .engine import FWD_ROUTE_ONLY

        self.engine.forward(layer, x2, gate_w, flags=FWD_ROUTE_ONLY)

    def row_elems(self):
        return self.engine.ep_row_elems()

    def pack(self, x2, send, counts, cap_rows):
        self.engine.ep_pack(x2, send, counts, cap_rows)

    def expert_ffn(self, layer, recv, y, cap_rows):
        self.engine.ep_expert_ffn(layer, recv, y, cap_rows)

    def combine(self, x2, ret, out, cap_rows):
        self.engine.ep_combine(x2, ret, out, cap_rows)


class ExpertParallelMoE:
    def __init__(self, ops, hidden: int, top_k: int, max_tokens: int, dtype: torch.dtype, device,
                 group: Optional[dist.ProcessGroup] = None):
        self.ops = ops
        self.group = group
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self.hidden, self.top_k = hidden, top_k
        self.cap_rows = max_tokens * top_k
        n = self.world * self.cap_rows
        mk = lambda *s, dt=dtype: torch.zeros(*s, dtype=dt, device=device)  # noqa: E731
        ld = ops.row_elems()  # H + 16-byte tail (expert id)
        self.send, self.recv = mk(n, ld), mk(n, ld)
        self.y, self.ret = mk(n, hidden), mk(n, hidden)

    def forward(self, layer: int, x: torch.Tensor, gate_w: torch.Tensor, out: Optional[torch.Tensor] = None):
        shape = x.shape
        x2 = x.reshape(-1, shape[-1])
        if x2.shape[0] * self.top_k > self.cap_rows:
            raise ValueError("more tokens than the exchange buffers were sized for")
        if out is None:
            out = torch.empty_like(x2)
        self.ops.route(layer, x2, gate_w)
        self.ops.pack(x2, self.send, None, self.cap_rows)
        # dispatch all-to-all: rows with their expert ids in the tail, equal splits of cap_rows per peer
        dist.all_to_all_single(self.recv, self.send, group=self.group)
        self.ops.expert_ffn(layer, self.recv, self.y, self.cap_rows)
        # combine all-to-all: expert outputs return to the rows' home rank, same row positions
        dist.all_to_all_single(self.ret, self.y, group=self.group)
        self.ops.combine(x2, self.ret, out, self.cap_rows)
        return out.reshape(shape)
