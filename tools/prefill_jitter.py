#!/usr/bin/env python3
"""Hunt the rare one-off 35-70 ms stall of prefill-sized forwards: <passes> passes of L layers x 512 tokens, one sync per
pass (as bench.py's prefill leg), per-pass wall time and the slowest host-side forward() call; run with
MOEINF_STALL_TRACE=5 to get the engine's own breakdown of any forward that held the host for more than 5 ms.
Measured (profiles/r03_host_jitter.txt): 300 passes of 4 layers, median 2.68 ms, max 3.62 ms, and 40 bench.py prefill passes
on another box all within 21.2-21.7 ms — the stalls come and go with the box (host noise), not with the engine.
usage: tools/prefill_jitter.py [passes] [layers] [tokens]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from moe_infinity_amd import MoEEngine, config as Cf  # noqa: E402
from oracle.synth import acts  # noqa: E402

passes = int(sys.argv[1]) if len(sys.argv) > 1 else 100
L = int(sys.argv[2]) if len(sys.argv) > 2 else 4
B = int(sys.argv[3]) if len(sys.argv) > 3 else 512
cfg = Cf.mixtral_8x7b(device_memory_ratio=0.5, max_tokens=B)
cfg.num_layers = L
eng = MoEEngine(cfg)
dev = torch.device("cuda:0")
off, siz, tot = eng.expert_layout(0)
for l in range(L):
    for e in range(cfg.num_experts):
        eng.register_expert(l, e, None)
        eng.expert_host_view(l, e).view(eng.dtype).copy_(torch.empty(tot // 2, dtype=eng.dtype, device=dev).normal_(0, 0.02))
    eng.prefetch(l, list(range(cfg.num_experts)))
eng.sync_copies()
gates = [(torch.randn(cfg.num_experts, cfg.hidden, device=dev) * 0.02).to(eng.gate_dtype) for _ in range(L)]
x = acts(B, cfg.hidden, eng.dtype, 10).to(dev)
out = torch.empty_like(x)
for l in range(L):
    eng.forward(l, x, gates[l], out=out)
torch.cuda.synchronize()
ts, worst_call = [], []
for p in range(passes):
    t0 = time.perf_counter()
    w = 0.0
    for l in range(L):
        c0 = time.perf_counter()
        eng.forward(l, x, gates[l], out=out)
        w = max(w, time.perf_counter() - c0)
    c0 = time.perf_counter()
    torch.cuda.synchronize()
    sync = time.perf_counter() - c0
    ts.append((time.perf_counter() - t0) * 1e3)
    worst_call.append((w * 1e3, sync * 1e3))
srt = sorted(ts)
print(f"{passes} passes of {L} layers x {B} tokens: median {srt[len(srt) // 2]:.3f} ms, p90 {srt[int(len(srt) * 0.9)]:.3f}, max {srt[-1]:.3f}")
for i, t in enumerate(ts):
    if t > 2.0 * srt[len(srt) // 2]:
        print(f"  pass {i}: {t:.2f} ms  (slowest forward() call {worst_call[i][0]:.2f} ms on the host, final sync {worst_call[i][1]:.2f} ms)")
eng.close()
