#!/usr/bin/env python3
"""A/B of batch-1 decode variants selected by environment knobs: the wall time of one MoE layer (every expert cached, the
sync-free path) for each `KEY=VAL,KEY=VAL` setting given on the command line, every setting in its own process, the whole
list run twice (A/B/A/B) so box drift shows.

    python tools/decode_ab.py deepseek-v2-lite base MOEINF_SR_LDS_KB=45 MOEINF_SR_LDS_KB=70,MOEINF_SR_U=4
"""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import sys, json, time, torch
sys.path.insert(0, %r)
from moe_infinity_amd import MoEEngine, config as Cf
wl, iters = sys.argv[1], int(sys.argv[2])
cfg = {"deepseek-v2-lite": Cf.deepseek_v2_lite, "mixtral-8x7b": Cf.mixtral_8x7b, "switch-base-8": Cf.switch_base_8}[wl](device_memory_ratio=0.75, max_tokens=1)
L = cfg.num_layers if wl != "mixtral-8x7b" else 8
cfg.num_layers = L
eng = MoEEngine(cfg); dev = torch.device("cuda:0")
off, siz, tot = eng.expert_layout(0)
es = 4 if eng.dtype == torch.float32 else 2
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
xs = [torch.randn(1, cfg.hidden, device=dev).to(eng.dtype) for i in range(8)]
out = torch.empty(1, cfg.hidden, dtype=eng.dtype, device=dev)
for i in range(3 * L): eng.forward(i %% L, xs[i %% 8], gates[i %% L], out=out)
torch.cuda.synchronize()
best = 1e9
for rep in range(5):
    t0 = time.perf_counter()
    for i in range(iters): eng.forward(i %% L, xs[i %% 8], gates[i %% L], out=out)
    torch.cuda.synchronize()
    best = min(best, (time.perf_counter() - t0) / iters * 1e6)
eng.set_profiling(True)
for i in range(iters): eng.forward(i %% L, xs[i %% 8], gates[i %% L], out=out)
p = eng.profile()
print("RESULT " + json.dumps(dict(layer_us=round(best, 2), ffn1_us=round(p["ffn1_ms"] * 1e3 / max(1, p["ffn1_launches"]), 2), ffn2_us=round(p["ffn2_ms"] * 1e3 / max(1, p["ffn2_launches"]), 2),
      route_us=round(p["route_ms"] * 1e3 / p["forwards"], 2))))
''' % ROOT
wl = sys.argv[1]
settings = sys.argv[2:] or ["base"]
for rnd in range(2):
    for s in settings:
        e = dict(os.environ)
        if s != "base":
            for kv in s.split(","):
                k, v = kv.split("=", 1)
                e[k] = v
        out = subprocess.run([sys.executable, "-c", CHILD, wl, "2600"], env=e, capture_output=True, text=True)
        r = [l for l in out.stdout.splitlines() if l.startswith("RESULT ")]
        print(f"{wl} round {rnd} {s:48s}", r[0][7:] if r else out.stderr[-800:], flush=True)
