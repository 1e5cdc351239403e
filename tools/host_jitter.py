#!/usr/bin/env python3
"""Are the rare one-off 35-70 ms stalls seen in timed forward loops (bench.py's prefill passes: 58.8 / 24.5 / 21.9 ms; a
3.5 ms/iteration blip in one of ~10 tools/ffn_sweep.py runs) pauses of Python's cyclic collector?  Per-iteration wall time of
a 512-token Mixtral layer (synchronised every iteration), the collector logging its own pauses; once as is, once after
gc.freeze().  Measured (profiles/r03_host_jitter.txt): no — 2 400 iterations, longest 0.91 ms, not one collection.  The
stalls are not reproduced by this loop; bench.py reports medians.  usage: tools/host_jitter.py [iters]"""
import gc
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from moe_infinity_amd import MoEEngine, config as Cf  # noqa: E402
from oracle.synth import acts  # noqa: E402

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 400
B, L = 512, 2
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
pauses = []
t_gc = [0.0]


def cb(phase, info):
    if phase == "start":
        t_gc[0] = time.perf_counter()
    else:
        pauses.append((info["generation"], (time.perf_counter() - t_gc[0]) * 1e3))


gc.callbacks.append(cb)


def run(tag):
    pauses.clear()
    ts = []
    for i in range(iters):
        t0 = time.perf_counter()
        eng.forward(i % L, x, gates[i % L], out=out)
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) * 1e3)
    ts_sorted = sorted(ts)
    print(f"{tag:12s} iters {iters}  median {ts_sorted[len(ts) // 2]:.3f} ms  p99 {ts_sorted[int(len(ts) * 0.99)]:.3f}  max {ts_sorted[-1]:.3f} ms at iteration {ts.index(ts_sorted[-1])}"
          f"  | gc pauses: {len(pauses)}, longest {max([p[1] for p in pauses], default=0):.1f} ms (generation {max(pauses, key=lambda p: p[1])[0] if pauses else '-'})")


for _ in range(6):
    eng.forward(0, x, gates[0], out=out)
torch.cuda.synchronize()
run("gc as is")
run("gc as is #2")
gc.collect()
gc.freeze()
run("gc.freeze()")
run("gc.freeze() #2")
eng.close()
