#!/usr/bin/env python3
"""One prefill-sized workload in THIS process (for rocprofv3 passes): tools/prefill_once.py <preset> <tokens> <layers> <iters>"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from moe_infinity_amd import MoEEngine, config as Cf  # noqa: E402

wl, B, L, iters = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
cfg = getattr(Cf, wl)(device_memory_ratio=0.5, max_tokens=B)
cfg.num_layers = L
eng = MoEEngine(cfg)
dev = torch.device("cuda:0")
off, siz, tot = eng.expert_layout(0)
es = 2 if eng.dtype == torch.bfloat16 else 4
for l in range(L):
    for e in range(cfg.num_experts):
        eng.register_expert(l, e, None)
        eng.expert_host_view(l, e).view(eng.dtype).copy_(torch.empty(tot // es, dtype=eng.dtype, device=dev).normal_(0, 0.02))
    if cfg.shared_inter:
        _, sizs, _ = eng.expert_layout(1)
        eng.register_shared(l, [torch.empty(s // es, dtype=eng.dtype).normal_(0, 0.02) for s in sizs])
    eng.prefetch(l, list(range(cfg.num_experts)))
eng.sync_copies()
gates = [(torch.randn(cfg.num_experts, cfg.hidden, device=dev) * 0.02).to(eng.gate_dtype) for _ in range(L)]
x = torch.randn(B, cfg.hidden, device=dev)
x = (x / x.pow(2).mean(-1, keepdim=True).sqrt()).to(eng.dtype)
out = torch.empty(B, cfg.hidden, dtype=eng.dtype, device=dev)
for i in range(iters):
    eng.forward(i % L, x, gates[i % L], out=out)
torch.cuda.synchronize()
eng.close()
