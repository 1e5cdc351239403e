#!/usr/bin/env python3
"""A yardstick for ffn_gemm_big: what the vendor library (hipBLASLt behind torch.matmul / torch.bmm) sustains on THIS chip for the
same bf16 products — one Mixtral-8x7B layer at 4 096 tokens (8 experts x 1 024 rows: gate/up [1024 x 4096] x [4096 x 14336] twice,
down [1024 x 14336] x [14336 x 4096]) as a batched GEMM, without any of the MoE work around it (no gather, no SiLU*mul, no ragged
counts) — and a square 8192^3 product.  Prints TFLOP/s; the dense bf16 peak at the nominal 2.4 GHz is 2 500.
(torch.bmm over a TRANSPOSED VIEW of the weights faulted on this stack — memory access fault in the library — so the nn.Linear
layout is measured with F.linear per expert.)"""
import time
import torch

dev = torch.device("cuda:0")
def bench(fn, flops, name, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    print(f"{name:70s} {dt * 1e6:9.1f} us  {flops / dt / 1e12:8.1f} TFLOP/s", flush=True)

E, T, H, F = 8, 1024, 4096, 14336
x = torch.randn(E, T, H, device=dev, dtype=torch.bfloat16)
w13 = torch.randn(E, H, 2 * F, device=dev, dtype=torch.bfloat16) * 0.02
w13t = torch.randn(E, 2 * F, H, device=dev, dtype=torch.bfloat16) * 0.02   # nn.Linear layout [out, in]
h = torch.randn(E, T, F, device=dev, dtype=torch.bfloat16)
w2 = torch.randn(E, F, H, device=dev, dtype=torch.bfloat16) * 0.02
w2t = torch.randn(E, H, F, device=dev, dtype=torch.bfloat16) * 0.02
bench(lambda: torch.bmm(x, w13), 2 * E * T * H * 2 * F, "gate+up as one batched GEMM [8 x 1024 x 4096] x [8 x 4096 x 28672] (NN)")
def lin_loop(a_, w_):
    for e in range(E):
        torch.nn.functional.linear(a_[e], w_[e])
bench(lambda: lin_loop(x, w13t), 2 * E * T * H * 2 * F, "  ... weights in the nn.Linear layout, F.linear per expert (8 launches)")
bench(lambda: torch.bmm(h, w2), 2 * E * T * F * H, "down as one batched GEMM [8 x 1024 x 14336] x [8 x 14336 x 4096] (NN)")
bench(lambda: lin_loop(h, w2t), 2 * E * T * F * H, "  ... weights in the nn.Linear layout, F.linear per expert (8 launches)")
a = torch.randn(8192, 8192, device=dev, dtype=torch.bfloat16)
b = torch.randn(8192, 8192, device=dev, dtype=torch.bfloat16)
bench(lambda: a @ b, 2 * 8192 ** 3, "square 8192^3 (NN)")
bench(lambda: torch.nn.functional.linear(a, b), 2 * 8192 ** 3, "square 8192^3, F.linear (NT)")
