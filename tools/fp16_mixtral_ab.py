#!/usr/bin/env python3
"""Mixtral-8x7B batch-1 decode with fp16 (or bf16) experts: ms per token over all 32 MoE layers.  MOEINF_DEC1_PAIR=0 is the
arrival-counter form of stage 2 (what fp16 ran before the pair kernel was instantiated for half_t).  One process per setting."""
import os, sys, time, torch
sys.path.insert(0, __file__.rsplit("/", 2)[0])
from moe_infinity_amd import MoEEngine, config as Cf

name = sys.argv[1] if len(sys.argv) > 1 else "fp16"
cfg = Cf.mixtral_8x7b(dtype=Cf.DTYPE_F16 if name == "fp16" else Cf.DTYPE_BF16, device_memory_ratio=0.75, max_tokens=1)
L = cfg.num_layers
eng = MoEEngine(cfg); dev = torch.device("cuda:0")
_, _, tot = eng.expert_layout(0)
for l in range(L):
    for e in range(cfg.num_experts):
        eng.register_expert(l, e, None)
        eng.expert_host_view(l, e).view(eng.dtype).copy_(torch.empty(tot // 2, dtype=eng.dtype, device=dev).normal_(0, 0.02))
    eng.prefetch(l, list(range(cfg.num_experts)))
eng.sync_copies()
gates = [(torch.randn(cfg.num_experts, cfg.hidden, device=dev) * 0.02).to(eng.gate_dtype) for _ in range(L)]
xs = [torch.randn(1, cfg.hidden, device=dev).to(eng.dtype) for i in range(8)]
out = torch.empty(1, cfg.hidden, dtype=eng.dtype, device=dev)
for i in range(3 * L): eng.forward(i % L, xs[i % 8], gates[i % L], out=out)
torch.cuda.synchronize()
res = []
for _ in range(3):
    t0 = time.perf_counter()
    for i in range(20 * L): eng.forward(i % L, xs[i % 8], gates[i % L], out=out)
    torch.cuda.synchronize()
    res.append((time.perf_counter() - t0) / 20 * 1e3)
print(f"dtype {name} MOEINF_DEC1_PAIR={os.environ.get('MOEINF_DEC1_PAIR', 'unset')}: ms/token " + " ".join(f"{r:.4f}" for r in res))
