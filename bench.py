#!/usr/bin/env python3
"""bench.py — decode-step benchmark of the expert-offload hot path on MI355X.

Contract (driver):  python bench.py --gpus N --steps K --warmup W
  N>1 is launched by the driver with torch.distributed.run (one rank per GPU, RCCL).
  Prints ONE JSON line on rank 0.

A "step" = one decode step of the hot path: one batch of B new tokens (B per rank) through ALL MoE
layers of the model (router -> dispatch index -> expert residency -> grouped expert FFN -> combine per
layer).  Attention/dense layers are outside the path (they stay on stock PyTorch in the reference
too) and are not in the timed region.  Inputs (per-layer activations, gate weights) are resident in
HBM before the timed region; expert weights live in the engine's pinned host arena and are cached
in HBM under device_memory_ratio exactly as the reference does.

Default workload (N=1): the configuration BASELINE.json's metric is quoted on — Mixtral-8x7B shapes
(32 MoE layers x 8 experts, H=4096 F=14336, top-2, bf16; 84 GiB of experts), device_memory_ratio
0.75, batch 1, synthetic weights N(0, 0.02^2) and RMS-normalised activations.
"""
import argparse
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8 TB/s (achievable ~6.3 TB/s)

WORKLOADS = {
    # name: (config factory name, family, human label)
    "mixtral-8x7b": ("mixtral_8x7b", "mixtral", "Mixtral-8x7B"),
    "deepseek-v2-lite": ("deepseek_v2_lite", "deepseek", "DeepSeek-V2-Lite"),
    "switch-base-8": ("switch_base_8", "switch", "Switch-base-8"),
    "nllb-moe-54b": ("nllb_moe_54b", "nllb", "NLLB-MoE-54B"),
}


def log(*a):
    if int(os.environ.get("RANK", "0")) == 0:
        print("[bench]", *a, file=sys.stderr, flush=True)


def fill_experts(eng, cfg, rank, world, dev, seed=1234):
    """Synthetic expert weights N(0, 0.02^2), generated on the GPU and copied into the engine's
    pinned host arena (the authoritative host tier)."""
    g = torch.Generator(device=dev)
    off, siz, tot = eng.expert_layout(0)
    dt = eng.dtype
    es = 2 if dt == torch.bfloat16 else 4
    t0 = time.time()
    n = 0
    for l in range(cfg.num_layers):
        for e in range(cfg.num_experts):
            if e % world != rank:
                continue
            eng.register_expert(l, e, None)
            g.manual_seed(seed + l * 1000 + e)
            blob = torch.empty(tot // es, dtype=dt, device=dev)
            blob.normal_(0.0, 0.02, generator=g)
            host = eng.expert_host_view(l, e).view(dt)
            host.copy_(blob)
            n += 1
    shared_host = {}
    if cfg.shared_inter:
        offs, sizs, tots = eng.expert_layout(1)
        for l in range(cfg.num_layers):
            g.manual_seed(seed + l * 1000 + 999)
            parts = []
            for s in sizs:
                t = torch.empty(s // es, dtype=dt, device=dev)
                t.normal_(0.0, 0.02, generator=g)
                parts.append(t.cpu())
            eng.register_shared(l, parts)
            shared_host[l] = parts
    torch.cuda.synchronize(dev)
    log(f"filled {n} experts ({n * tot / 2**30:.1f} GiB pinned) in {time.time() - t0:.1f}s")
    return shared_host


def host_expert_tensors(eng, cfg, layer, expert):
    """Zero-copy torch views (CPU) of one expert's tensors inside the pinned arena, reference blob order."""
    off, siz, _ = eng.expert_layout(0)
    raw = eng.expert_host_view(layer, expert)
    H, F = cfg.hidden, cfg.inter
    from moe_infinity_amd import config as Cf

    if cfg.expert_type in (Cf.EXPERT_MIXTRAL,):
        shapes = [(F, H), (H, F), (F, H)]
    elif cfg.expert_type == Cf.EXPERT_DEEPSEEK:
        shapes = [(F, H), (F, H), (H, F)]
    elif cfg.expert_type == Cf.EXPERT_SWITCH:
        shapes = [(F, H), (H, F)]
    else:
        shapes = [(F, H), (F,), (H, F), (H,)]
    return [raw[o:o + s].view(eng.dtype).reshape(sh) for o, s, sh in zip(off, siz, shapes)]


def main():
    # Native libraries (RCCL prints a version banner, HIP/driver warnings) write to the C stdout; keep the
    # process's real stdout for the ONE JSON line only.
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="mixtral-8x7b", choices=sorted(WORKLOADS))
    ap.add_argument("--batch", type=int, default=1, help="decode batch (new tokens per step) per rank")
    ap.add_argument("--ratio", type=float, default=0.75, help="device_memory_ratio")
    ap.add_argument("--budget-gib", type=float, default=0.0, help="explicit expert-cache budget (miss-heavy runs)")
    ap.add_argument("--layers", type=int, default=0, help="override the number of MoE layers (0 = the model's)")
    ap.add_argument("--policy", default="lfu_incache", choices=["lfu_incache", "lru"])
    ap.add_argument("--prompt", type=int, default=512, help="prefill length run once before decoding (examples/interface_example.py protocol); 0 = skip")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--force-ep", action="store_true", help="exercise the expert-parallel path even with one rank (testing)")
    ap.add_argument("--cpu-sample-layers", type=int, default=4)
    ap.add_argument("--cpu-sample-steps", type=int, default=3)
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus N>1 must be launched with torch.distributed.run (one rank per GPU)")
    import torch.distributed as dist

    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    use_ep = world > 1 or args.force_ep
    if use_ep:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group("nccl", device_id=dev, rank=rank, world_size=world)

    import __graft_entry__ as entry

    if rank == 0:
        entry.build()
    if world > 1:
        dist.barrier()
    from moe_infinity_amd import MoEEngine
    from moe_infinity_amd import config as Cf
    from moe_infinity_amd.ep import ExpertParallelMoE, HipEpOps

    factory, family, label = WORKLOADS[args.workload]
    B = args.batch
    cfg = getattr(Cf, factory)(device_id=local_rank, device_memory_ratio=args.ratio,
                               device_memory_bytes=int(args.budget_gib * 2**30),
                               policy=Cf.POLICY_LRU if args.policy == "lru" else Cf.POLICY_LFU_INCACHE,
                               ep_rank=rank, ep_size=world, max_tokens=max(B * world, B * args.prompt if world == 1 else 0))
    if args.layers:
        cfg.num_layers = args.layers
    L, E, K, H = cfg.num_layers, cfg.num_experts, cfg.top_k, cfg.hidden
    eng = MoEEngine(cfg)
    shared_host = fill_experts(eng, cfg, rank, world, dev)
    dt = eng.dtype
    gdt = eng.gate_dtype
    gg = torch.Generator(device=dev)
    gates = []
    for l in range(L):
        gg.manual_seed(4321 + l)  # identical on every rank: the router is replicated
        gstd = 0.02 if family in ("mixtral", "deepseek") else 0.5
        gates.append((torch.randn(E, H, generator=gg, device=dev) * gstd).to(gdt))

    from oracle.synth import acts  # synthetic activation protocol (seed 2024+layer, RMS-normalised rows)

    nsteps = args.warmup + args.steps
    xs = [[acts(B, H, dt, 2024 + l + 1000 * s + 100000 * rank).to(dev) for l in range(L)] for s in range(nsteps)]
    out = torch.empty(B, H, dtype=dt, device=dev)
    batch_rows = B if family == "switch" else 1

    if use_ep:
        ep = ExpertParallelMoE(HipEpOps(eng), H, K, B, dt, dev)

        def layer_fwd(l, x):
            ep.forward(l, x, gates[l], out=out)
    else:
        def layer_fwd(l, x):
            eng.forward(l, x, gates[l], batch_rows=batch_rows, out=out)

    def run_steps(s0, n):
        for s in range(s0, s0 + n):
            for l in range(L):
                layer_fwd(l, xs[s][l])

    def fence():
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    # ---- warm-up.  A real decode phase follows a prefill that has already touched (nearly) every
    # expert, so the cache starts warm: stream every owned expert that fits the budget through the
    # prefetch path, then W decode steps; counters are reset at the prefill->decode boundary like
    # examples/interface_example.py:39 does.
    t0 = time.time()
    for l in range(L):
        eng.prefetch(l, [e for e in range(E) if e % world == rank])
    eng.sync_copies()
    log(f"cache warm ({eng.stats()['slots_used']} experts resident) in {time.time() - t0:.1f}s, h2d {eng.stats()['h2d_bytes'] / 2**30:.1f} GiB, copy-busy {eng.stats()['h2d_busy_ms']:.0f} ms")
    warm = eng.stats()
    # prefill of the prompt (B sequences x --prompt tokens) through every layer: exercises the large-T
    # path; timed separately, NOT part of `value`
    prefill_ms = None
    if args.prompt > 0 and not use_ep:
        xp = acts(B * args.prompt, H, dt, 777).to(dev)
        outp = torch.empty_like(xp)
        for l in range(L):  # untimed pass: makes every expert the prompt touches resident
            eng.forward(l, xp, gates[l], batch_rows=batch_rows, out=outp)
        torch.cuda.synchronize(dev)
        tp = time.perf_counter()
        for l in range(L):
            eng.forward(l, xp, gates[l], batch_rows=batch_rows, out=outp)
        torch.cuda.synchronize(dev)
        prefill_ms = (time.perf_counter() - tp) * 1e3
        del xp, outp
    run_steps(0, args.warmup)
    eng.sync_copies()
    fence()
    log(f"warm-up {args.warmup} steps in {time.time() - t0:.1f}s; stats {eng.stats()}")
    eng.clear_expert_cache_counts()
    eng.reset_stats()

    # ---- timed region: exactly K steps
    fence()
    t0 = time.perf_counter()
    run_steps(args.warmup, args.steps)
    fence()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    ms_per_step = elapsed * 1e3 / args.steps
    tokens_per_s = world * B * args.steps / elapsed
    st = eng.stats()

    # ---- same K steps again with per-kernel HIP events on the launch stream (roofline leg)
    # Expert-parallel runs: every rank repeats the steps (the collectives need all of them); the events bracket the
    # owner-side FFN launches of THIS rank, and rank 0 reports its own kernels.
    roof, kernels = None, {}
    p = None
    try:
        eng.set_profiling(True)
        fence()
        run_steps(args.warmup, args.steps)
        fence()
        p = eng.profile()
        eng.set_profiling(False)
    except Exception as ex:  # the measured line above must survive a failure of this extra leg in multi-rank runs
        if world == 1:
            raise
        log(f"roofline leg failed on rank {rank}: {ex!r}")
    if p is not None:

        def kstat(ms, launches, nbytes):
            if launches == 0 or ms <= 0:
                return None
            us = ms * 1e3 / launches
            gbs = nbytes / (ms * 1e-3) / 1e9
            return {"avg_launch_us": round(us, 3), "bytes_per_launch": int(nbytes // launches),
                    "achieved_GBps": round(gbs, 1), "frac_of_hbm_peak": round(gbs / HBM_PEAK_GBS, 4)}

        kernels = {"ffn_stage1": kstat(p["ffn1_ms"], p["ffn1_launches"], p["ffn1_bytes"]),
                   "ffn_stage2": kstat(p["ffn2_ms"], p["ffn2_launches"], p["ffn2_bytes"])}
        if use_ep:
            kernels["note"] = f"rank 0's owner-side FFN over the rows it received (experts e % {world} == 0)"
        else:
            kernels.update({
                   "route(gate+topk+index)": kstat(p["route_ms"], p["forwards"], p["route_bytes"]),
                   "combine": kstat(p["combine_ms"], p["forwards"], p["combine_bytes"]),
                   # decode-sized Mixtral/DeepSeek steps: the combine runs in the epilogue of FFN stage 2, the
                   # "combine" interval above is then an empty event-to-event interval (= the events' own cost)
                   "combine_fused_into_ffn_stage2": bool(B <= 16 and family in ("mixtral", "deepseek")
                                                         and os.environ.get("MOEINF_FUSE_COMBINE", "1") != "0"),
                   "host_wait_ms_per_layer": round(p["host_wait_ms"] / max(1, p["forwards"]), 4)})
        k1 = kernels["ffn_stage1"]
        # HBM traffic per launch of the dominant kernel from the committed PMC passes (separate
        # rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE runs of this same workload; tools/pmc_summary.py)
        traffic = None
        pmc = os.path.join(ROOT, "profiles", "r01_pmc_traffic_mixtral8x7b.json")
        if args.workload == "mixtral-8x7b" and B == 1 and not use_ep and os.path.exists(pmc):
            for name, v in json.load(open(pmc))["kernels"].items():
                if "ffn_rows_kernel<unsigned short, 2" in name:
                    traffic = v["hbm_bytes"]
        if k1:
            roof = {"bound": "hbm", "kernel": "ffn_rows_kernel stage 1 (gate/up rows of the active experts, fused gather + act)"
                                              + (f"; rank 0 of {world}, owner-side launches" if use_ep else ""),
                    "achieved": k1["achieved_GBps"], "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": k1["frac_of_hbm_peak"],
                    "traffic": traffic, "avg_launch_us": k1["avg_launch_us"], "bytes_per_launch": k1["bytes_per_launch"]}

    # ---- CPU baseline + full-size parity: the oracle on a bounded sample of the same workload
    cpu = None
    parity = None
    if rank == 0 and world == 1 and not use_ep and not args.no_cpu_baseline:
        from oracle import moe_ref as R

        ncores = os.cpu_count() or 1
        torch.set_num_threads(ncores)
        ls = list(range(min(args.cpu_sample_layers, L)))
        ss = list(range(args.warmup, args.warmup + min(args.cpu_sample_steps, args.steps)))

        def oracle_layer(l, x_cpu):
            gate = gates[l].cpu()
            experts = [host_expert_tensors(eng, cfg, l, e) for e in range(E)]
            if family == "mixtral":
                return R.block_mixtral(x_cpu[None], gate, experts, top_k=K)
            if family == "deepseek":
                Fs, Hh = cfg.shared_inter, cfg.hidden
                sp = shared_host[l]
                shared = [sp[0].view(Fs, Hh), sp[1].view(Fs, Hh), sp[2].view(Hh, Fs)]
                return R.block_deepseek(x_cpu[None], gate, experts, K, shared=shared, norm_topk_prob=bool(cfg.norm_topk_prob),
                                        routed_scaling_factor=cfg.routed_scaling_factor)
            if family == "switch":
                return R.block_switch(x_cpu[None], gate, experts, expert_capacity=cfg.expert_capacity)
            return R.block_nllb(x_cpu[None], gate, experts)

        if True:
            oracle_layer(ls[0], xs[ss[0]][ls[0]].cpu())  # warm the CPU path
            t0 = time.perf_counter()
            refs = {}
            for s in ss:
                for l in ls:
                    refs[(s, l)] = oracle_layer(l, xs[s][l].cpu())
            cpu_s = time.perf_counter() - t0
            layer_steps = len(ss) * len(ls)
            cpu_ms_per_token = cpu_s * 1e3 / layer_steps * L / B  # extrapolated to all L layers
            cpu = {"value": round(1e3 / cpu_ms_per_token, 4), "unit": "tokens/s", "cores": ncores, "kind": "port",
                   "ms_per_token": round(cpu_ms_per_token, 2),
                   "sample": f"{len(ss)} decode steps x layers {ls[0]}..{ls[-1]} of the same workload "
                             f"({layer_steps} MoE-layer passes, {cpu_s:.1f}s), extrapolated x{L}/{len(ls)} layers; "
                             f"torch CPU ops, {ncores} threads"}
            # parity of the full-size GPU path on the sampled (step, layer) pairs
            worst, exact = 0.0, True
            for (s, l), ref in refs.items():
                o = eng.forward(l, xs[s][l], gates[l], batch_rows=batch_rows).float().cpu()
                r = eng.routing()
                if family == "mixtral":
                    exact &= bool((torch.from_numpy(r["topk_idx"]).long() == ref.topk_idx).all())
                elif family == "deepseek":
                    exact &= all(sorted(a.tolist()) == sorted(b.tolist()) for a, b in zip(r["topk_idx"], ref.topk_idx.numpy()))
                want = ref.out[0].float()
                tol = torch.maximum(torch.maximum(want.abs(), o.abs()), want.abs().mean()) * (2.0 ** -7 if dt == torch.bfloat16 else 2e-5)
                worst = max(worst, float(((o - want).abs() / tol).max()))
            parity = {"routing_bit_exact": exact, "max_err_ulps_of_dtype": round(worst, 3), "pairs_checked": len(refs)}

    if rank == 0:
        line = {
            "metric": f"decode tokens/s through all MoE layers (expert-offload hot path), {label}, device_memory_ratio={args.ratio}",
            "value": round(tokens_per_s, 3), "unit": "tokens/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "bf16" if dt == torch.bfloat16 else "f32", "data": "synthetic",
            "config": {"workload": f"{label} MoE layers: L={L} E={E} K={K} H={H} F={cfg.inter}"
                                   + (f" +shared F={cfg.shared_inter}" if cfg.shared_inter else "")
                                   + f", decode batch {B}/rank, device_memory_ratio={args.ratio}"
                                   + (f", expert-cache budget {args.budget_gib} GiB" if args.budget_gib else ""),
                       "parallelism": f"ep{world}" if use_ep else "single", "per_token_decode_latency_ms": round(ms_per_step, 4),
                       "cache_policy": args.policy},
            "prefill": None if prefill_ms is None else {"tokens": B * args.prompt, "ms_all_layers": round(prefill_ms, 2),
                                                        "tokens_per_s": round(B * args.prompt / prefill_ms * 1e3, 1)},
            "roofline": roof,
            "cpu_baseline": cpu,
            "kernels": kernels,
            "prefetch_stream": None if warm["h2d_busy_ms"] <= 0 else {
                "what": "cache warm-up: every owned expert streamed host(pinned)->HBM on the prefetch stream (hipMemcpyAsync + device re-tile)",
                "GiB": round(warm["h2d_bytes"] / 2**30, 2), "busy_ms": round(warm["h2d_busy_ms"], 1),
                "GBps": round(warm["h2d_bytes"] / warm["h2d_busy_ms"] / 1e6, 2),
                "frac_of_pcie5_x16_63GBps": round(warm["h2d_bytes"] / warm["h2d_busy_ms"] / 1e6 / 63.0, 3),
                "frac_of_hbm_peak": round(warm["h2d_bytes"] / warm["h2d_busy_ms"] / 1e6 / HBM_PEAK_GBS, 4)},
            "timed_region_h2d": {"bytes": st["h2d_bytes"], "busy_ms": round(st["h2d_busy_ms"], 2),
                                 "GBps": round(st["h2d_bytes"] / st["h2d_busy_ms"] / 1e6, 2) if st["h2d_busy_ms"] > 0 else None,
                                 "exposed_wait_ms": round(st["exposed_wait_ms"], 2),
                                 "overlap": None if st["h2d_busy_ms"] <= 0 else round(max(0.0, 1.0 - st["exposed_wait_ms"] / st["h2d_busy_ms"]), 4),
                                 "hit_rate": round(st["expert_hits"] / max(1, st["expert_hits"] + st["expert_misses"]), 4)},
            "cache": {k: st[k] for k in ("expert_hits", "expert_misses", "evictions", "h2d_bytes", "slots_total", "slots_used", "slot_bytes", "host_arena_bytes")},
            "parity": parity,
        }
        sys.stdout.flush()
        os.write(real_stdout, (json.dumps(line) + "\n").encode())
    eng.close()
    if use_ep:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
