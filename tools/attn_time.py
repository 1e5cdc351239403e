"""How long is the NON-MoE part of a decoder layer at decode time on this GPU?  (Input of tools/prefetch_study.py: a
speculative expert copy can only hide under what runs between two MoE layers.)

Stock PyTorch-ROCm ops, bf16, random weights: rmsnorm -> qkv projection -> scaled_dot_product_attention over a KV cache of
`ctx` tokens -> output projection -> rmsnorm, for Mixtral-8x7B (32 query heads, 8 KV heads, head 128, H 4096) and a
DeepSeek-V2-Lite-shaped layer (16 heads, qk 192 / v 128 after up-projection, H 2048; MLA's compressed cache is expanded
as HF's eager implementation does).  Prints one JSON line per (model, batch, context): microseconds per layer.
"""
import json
import sys

import torch
import torch.nn.functional as F


def layer_fn(model, B, ctx, dev):
    dt = torch.bfloat16
    if model == "mixtral-8x7b":
        H, nq, nkv, dqk, dv = 4096, 32, 8, 128, 128
    else:
        H, nq, nkv, dqk, dv = 2048, 16, 16, 192, 128
    wq = torch.randn(nq * dqk, H, device=dev, dtype=dt) * 0.02
    wk = torch.randn(nkv * dqk, H, device=dev, dtype=dt) * 0.02
    wv = torch.randn(nkv * dv, H, device=dev, dtype=dt) * 0.02
    wo = torch.randn(H, nq * dv, device=dev, dtype=dt) * 0.02
    g1 = torch.ones(H, device=dev, dtype=dt)
    kc = torch.randn(B, nkv, ctx, dqk, device=dev, dtype=dt)
    vc = torch.randn(B, nkv, ctx, dv, device=dev, dtype=dt)
    x = torch.randn(B, 1, H, device=dev, dtype=dt)

    def rms(h, g):
        return (h.float() * torch.rsqrt(h.float().pow(2).mean(-1, keepdim=True) + 1e-5)).to(dt) * g

    def f():
        h = rms(x, g1)
        q = F.linear(h, wq).view(B, 1, nq, dqk).transpose(1, 2)
        k = F.linear(h, wk).view(B, 1, nkv, dqk).transpose(1, 2)
        v = F.linear(h, wv).view(B, 1, nkv, dv).transpose(1, 2)
        kk = torch.cat([kc[:, :, 1:], k], 2)
        vv = torch.cat([vc[:, :, 1:], v], 2)
        if dqk != dv:  # SDPA wants equal head sizes: pad v (what HF's DeepSeek eager path effectively pays)
            vv = F.pad(vv, (0, dqk - dv))
        o = F.scaled_dot_product_attention(q, kk, vv, enable_gqa=(nq != nkv))
        o = o[..., :dv].transpose(1, 2).reshape(B, 1, nq * dv)
        y = x + F.linear(o, wo)
        return rms(y, g1)

    return f


def main():
    dev = torch.device("cuda", 0)
    out = []
    for model in ("mixtral-8x7b", "deepseek-v2-lite"):
        for B in (1, 8):
            for ctx in (2048, 8192):
                f = layer_fn(model, B, ctx, dev)
                for _ in range(10):
                    f()
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                n = 100
                e0.record()
                for _ in range(n):
                    f()
                e1.record()
                torch.cuda.synchronize()
                rec = {"model": model, "batch": B, "context": ctx, "attention_block_us_per_layer": round(e0.elapsed_time(e1) * 1e3 / n, 1),
                       "ops": "stock PyTorch-ROCm bf16: rmsnorm, q/k/v linear, cache append (cat), SDPA, o linear, residual, rmsnorm"}
                print(json.dumps(rec), flush=True)
                out.append(rec)
    return 0


if __name__ == "__main__":
    sys.exit(main())
