#!/usr/bin/env python3
"""Sweep the FFN kernels' tuning knobs (env MOEINF_FFN_* / MOEINF_GEMM_*) on real model
shapes; one subprocess per setting so the static env lookups are fresh.  Prints per-stage us and GB/s.
usage: ffn_sweep.py <preset>:<tokens>:<layers> ...   e.g.  mixtral_8x7b:512:2 nllb_moe_54b:2048:1"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import sys, os, json, torch
sys.path.insert(0, %r)
from moe_infinity_amd import MoEEngine, config as Cf
from oracle.synth import acts
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
gstd = 0.02 if cfg.router_kind in (0, 1) else 0.5
gates = [(torch.randn(cfg.num_experts, cfg.hidden, device=dev) * gstd).to(eng.gate_dtype) for _ in range(L)]
xs = [acts(B, cfg.hidden, eng.dtype, 10 + i).to(dev) for i in range(8)]
out = torch.empty(B, cfg.hidden, dtype=eng.dtype, device=dev)
br = B if cfg.router_kind == 2 else 1
for i in range(3 * L):
    eng.forward(i %% L, xs[i %% 8], gates[i %% L], batch_rows=br, out=out)
torch.cuda.synchronize()
eng.set_profiling(True)
import time
t0 = time.perf_counter()
for i in range(iters):
    eng.forward(i %% L, xs[i %% 8], gates[i %% L], batch_rows=br, out=out)
torch.cuda.synchronize()
wall = (time.perf_counter() - t0) / iters * 1e6
p = eng.profile()
def st(ms, n, b):
    return (ms * 1e3 / max(1, n), b / max(ms, 1e-9) / 1e6)
r = dict(wall_us=round(wall, 1), ffn1=st(p["ffn1_ms"], p["ffn1_launches"], p["ffn1_bytes"]), ffn2=st(p["ffn2_ms"], p["ffn2_launches"], p["ffn2_bytes"]),
         route_us=p["route_ms"] * 1e3 / p["forwards"], combine_us=p["combine_ms"] * 1e3 / p["forwards"])
print("RESULT " + json.dumps(r))
''' % ROOT


def run(wl, B, L, iters, env):
    e = dict(os.environ)
    e.update({k: str(v) for k, v in env.items()})
    out = subprocess.run([sys.executable, "-c", CHILD, wl, str(B), str(L), str(iters)], env=e, capture_output=True, text=True)
    for line in out.stdout.splitlines():
        if line.startswith("RESULT "):
            return json.loads(line[7:])
    return {"error": (out.stderr or out.stdout)[-400:]}


if __name__ == "__main__":
    wls = sys.argv[1:] or ["mixtral_8x7b:1:4", "deepseek_v2_lite:1:4"]
    for spec in wls:
        wl, B, L = spec.split(":")
        # default sweep: the shipped choice vs the main alternatives (see DESIGN.md section 4.3 for every knob)
        envs = [{}] if int(B) <= 16 else [{}, {"MOEINF_GEMM_RING2": 0}, {"MOEINF_RING2_TAIL": 0}, {"MOEINF_GEMM_BIG": 0}]
        if os.environ.get("SWEEP_ENVS"):
            envs = [dict(kv.split("=") for kv in e.split(",") if kv) for e in os.environ["SWEEP_ENVS"].split(";")]
        for env in envs:
            r = run(wl, int(B), int(L), 200 if int(B) <= 16 else 20, env)
            if "error" in r:
                print(spec, env, "ERROR", r["error"])
                continue
            print(f"{spec:24s} {str(env):70s} ffn1 {r['ffn1'][0]:7.1f}us {r['ffn1'][1]:7.0f}GB/s | ffn2 {r['ffn2'][0]:7.1f}us {r['ffn2'][1]:7.0f}GB/s | route {r['route_us']:5.1f} comb {r['combine_us']:5.1f} wall {r['wall_us']:7.1f}", flush=True)
