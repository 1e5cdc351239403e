#!/usr/bin/env python3
"""DeepSeek-V2-Lite decode experiments (one MoE layer loop, B=1): what would hiding the shared expert under the router
buy?  Variant 'noshared' runs the same routed experts without the shared expert."""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import sys, json, time, torch
sys.path.insert(0, %r)
from moe_infinity_amd import MoEEngine, config as Cf
variant, L, iters = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
kw = {}
cfg = Cf.deepseek_v2_lite(device_memory_ratio=0.5, max_tokens=1)
if variant == "noshared": cfg.shared_inter = 0
cfg.num_layers = L
eng = MoEEngine(cfg); dev = torch.device("cuda:0")
off, siz, tot = eng.expert_layout(0)
for l in range(L):
    for e in range(cfg.num_experts):
        eng.register_expert(l, e, None)
        eng.expert_host_view(l, e).view(eng.dtype).copy_(torch.empty(tot // 2, dtype=eng.dtype, device=dev).normal_(0, 0.02))
    if cfg.shared_inter:
        _, sizs, _ = eng.expert_layout(1)
        eng.register_shared(l, [torch.empty(s // 2, dtype=eng.dtype).normal_(0, 0.02) for s in sizs])
    eng.prefetch(l, list(range(cfg.num_experts)))
eng.sync_copies()
gates = [(torch.randn(cfg.num_experts, cfg.hidden, device=dev) * 0.02).to(eng.gate_dtype) for _ in range(L)]
xs = [torch.randn(1, cfg.hidden, device=dev).to(eng.dtype) for i in range(8)]
out = torch.empty(1, cfg.hidden, dtype=eng.dtype, device=dev)
for i in range(3 * L): eng.forward(i %% L, xs[i %% 8], gates[i %% L], out=out)
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(iters): eng.forward(i %% L, xs[i %% 8], gates[i %% L], out=out)
torch.cuda.synchronize()
wall = (time.perf_counter() - t0) / iters * 1e6
eng.set_profiling(True)
for i in range(iters): eng.forward(i %% L, xs[i %% 8], gates[i %% L], out=out)
p = eng.profile()
print("RESULT " + json.dumps(dict(wall_us=round(wall, 2), ffn1_us=round(p["ffn1_ms"] * 1e3 / p["ffn1_launches"], 2), ffn2_us=round(p["ffn2_ms"] * 1e3 / max(1, p["ffn2_launches"]), 2),
      route_us=round(p["route_ms"] * 1e3 / p["forwards"], 2), ffn1_MB=p["ffn1_bytes"] / p["ffn1_launches"] / 1e6, ffn2_MB=p["ffn2_bytes"] / max(1, p["ffn2_launches"]) / 1e6)))
''' % ROOT
for variant in sys.argv[1:] or ["base", "noshared"]:
    envs = [{}]
    for env in envs:
        e = dict(os.environ); e.update(env)
        out = subprocess.run([sys.executable, "-c", CHILD, variant, "26", "2600"], env=e, capture_output=True, text=True)
        r = [l for l in out.stdout.splitlines() if l.startswith("RESULT ")]
        print(variant, env, r[0][7:] if r else out.stderr[-500:], flush=True)
