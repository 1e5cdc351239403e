"""Host-side cost of one MoEEngine.forward (Python + ctypes + launches) against the GPU time of a layer, batch 1."""
import sys, time, torch
sys.path.insert(0, __file__.rsplit("/", 2)[0])
import __graft_entry__ as g; g.build()
from moe_infinity_amd import MoEEngine, config as Cf
for wl in ("switch_base_8", "deepseek_v2_lite", "mixtral_8x7b"):
    cfg = getattr(Cf, wl)(device_memory_ratio=0.75, max_tokens=1)
    L = cfg.num_layers = 8
    es = 4 if cfg.dtype == Cf.DTYPE_F32 else 2
    eng = MoEEngine(cfg); dev = torch.device("cuda:0")
    off, siz, tot = eng.expert_layout(0)
    for l in range(L):
        for e in range(cfg.num_experts):
            eng.register_expert(l, e, None)
            eng.expert_host_view(l, e).view(eng.dtype).copy_(torch.empty(tot // es, dtype=eng.dtype, device=dev).normal_(0, 0.02))
        if cfg.shared_inter:
            _, sizs, _ = eng.expert_layout(1)
            eng.register_shared(l, [torch.empty(s // es, dtype=eng.dtype).normal_(0, 0.02) for s in sizs])
        eng.prefetch(l, list(range(cfg.num_experts)))
    eng.sync_copies()
    gates = [(torch.randn(cfg.num_experts, cfg.hidden, device=dev) * (0.5 if wl.startswith("switch") else 0.02)).to(eng.gate_dtype) for _ in range(L)]
    x = torch.randn(1, cfg.hidden, device=dev).to(eng.dtype); out = torch.empty_like(x)
    for i in range(200): eng.forward(i % L, x, gates[i % L], out=out)
    torch.cuda.synchronize()
    N = 4000
    t0 = time.perf_counter()
    for i in range(N): eng.forward(i % L, x, gates[i % L], out=out)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    # host cost alone: short bursts right after a sync (the launch queue is empty: no back-pressure)
    burst = []
    for _ in range(50):
        torch.cuda.synchronize()
        tb = time.perf_counter()
        for i in range(16): eng.forward(i % L, x, gates[i % L], out=out)
        burst.append((time.perf_counter() - tb) / 16 * 1e6)
    torch.cuda.synchronize()
    burst.sort()
    print(wl, "host-only us/layer (median of 50 bursts of 16)", round(burst[25], 2), "min", round(burst[0], 2))
    print(wl, "host enqueue us/layer", round((t1 - t0) / N * 1e6, 2), "total us/layer", round((t2 - t0) / N * 1e6, 2), flush=True)
    eng.close()
