#!/usr/bin/env python3
"""DeepSeek-V2-Lite batch-1 decode with fp16 experts: ms per token over all 26 MoE layers, with the shared expert hidden under
the router (MOEINF_HIDE_SHARED unset: moe_front1 + ffn2_decode1 since round 5) or behind it (MOEINF_HIDE_SHARED=0: the generic
path fp16 took before).  One process per setting (the knob is read once)."""
import sys, time, torch
sys.path.insert(0, __file__.rsplit("/", 2)[0])
from moe_infinity_amd import MoEEngine, config as Cf

dt_id = Cf.DTYPE_F16 if (len(sys.argv) < 2 or sys.argv[1] == "fp16") else Cf.DTYPE_BF16
cfg = Cf.deepseek_v2_lite(dtype=dt_id, device_memory_ratio=0.5, max_tokens=1)
L = cfg.num_layers
eng = MoEEngine(cfg); dev = torch.device("cuda:0")
_, _, tot = eng.expert_layout(0)
for l in range(L):
    for e in range(cfg.num_experts):
        eng.register_expert(l, e, None)
        eng.expert_host_view(l, e).view(eng.dtype).copy_(torch.empty(tot // 2, dtype=eng.dtype, device=dev).normal_(0, 0.02))
    _, sizs, _ = eng.expert_layout(1)
    eng.register_shared(l, [torch.empty(s // 2, dtype=eng.dtype).normal_(0, 0.02) for s in sizs])
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
    for i in range(40 * L): eng.forward(i % L, xs[i % 8], gates[i % L], out=out)
    torch.cuda.synchronize()
    res.append((time.perf_counter() - t0) / 40 * 1e3)
import os
print(f"dtype {sys.argv[1] if len(sys.argv) > 1 else 'fp16'} MOEINF_HIDE_SHARED={os.environ.get('MOEINF_HIDE_SHARED', 'unset')}: ms/token " + " ".join(f"{r:.4f}" for r in res))
