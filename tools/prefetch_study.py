#!/usr/bin/env python3
"""Cache-miss regime study (BASELINE config "cache-miss-heavy"): expert cache smaller than the expert
set, an attention stand-in between MoE layers, and SEQUENCE-SPECIFIC skewed routing: every sequence
belongs to one of `--types` types, each type favours its own experts per layer (Zipf logit bias).  That
is the structure MoE-Infinity's activation-aware prefetch exploits: the history holds EAMs of earlier
sequences of every type, the nearest EAM identifies the running sequence's type after a few layers.
Policies:
  lfu            on-demand fetches only, LFU-in-cache eviction (what the reference runs)
  lru            on-demand fetches only, LRU eviction
  lfu+prefetch_all  the reference's prefetcher as written: enqueue EVERY predicted expert of later layers
  lfu+prefetch   same, but only experts predicted to take >= --min-share of their layer's activations
  lfu+prefetch+governor  same, with the engine's speculation governor on (moeinf_set_prefetch_governor(0.5, 16))
Prints one JSON object per policy: ms/token, hit rate, H2D GB/s, exposed copy wait, overlap.
Usage: python tools/prefetch_study.py [--workload mixtral_8x7b] [--layers 8] [--cache-frac 0.5] [--steps 40]
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="mixtral_8x7b")
    ap.add_argument("--layers", type=int, default=8)
    ap.add_argument("--cache-frac", type=float, default=0.5)
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--hist-seqs", type=int, default=8)
    ap.add_argument("--zipf", type=float, default=1.2)
    ap.add_argument("--attn-us", type=float, default=150.0)
    ap.add_argument("--max-prefetch", type=int, default=16)
    ap.add_argument("--min-share", type=float, default=0.15)
    ap.add_argument("--types", type=int, default=4)
    ap.add_argument("--seqs", type=int, default=8)
    ap.add_argument("--seq-len", type=int, default=10)
    ap.add_argument("--policies", default="", help="comma-separated subset of the policies")
    ap.add_argument("--batch", type=int, default=1, help="decode batch: row i of a step belongs to a sequence of type (seq + i) %% types; ONE trace entry per batch")
    ap.add_argument("--note", default="", help="free text copied into every result line (e.g. where --attn-us comes from)")
    args = ap.parse_args()

    from moe_infinity_amd import MoEEngine
    from moe_infinity_amd import config as Cf
    from moe_infinity_amd.engine import FWD_ROUTE_ONLY
    from moe_infinity_amd.memory import ExpertPredictor, ExpertPrefetcher, ExpertTracer
    from oracle.synth import acts

    dev = torch.device("cuda:0")
    L = args.layers
    results = []
    # attention stand-in: a bf16 matmul sized to ~attn_us on the compute stream
    n = 2048
    a = torch.randn(n, n, device=dev, dtype=torch.bfloat16)
    b = torch.randn(n, n, device=dev, dtype=torch.bfloat16)
    torch.cuda.synchronize()
    for _ in range(5):
        a @ b
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(50):
        a @ b
    torch.cuda.synchronize()
    one = (time.perf_counter() - t0) / 50 * 1e6
    reps = max(1, int(round(args.attn_us / one)))

    policies = ("lfu", "lru", "lfu+prefetch_all", "lfu+prefetch", "lfu+prefetch+governor",
                "lfu+engine_predictor_la1", "lfu+engine_predictor_la2+governor", "lfu+engine_predictor_la8+governor")
    if args.policies:
        policies = tuple(args.policies.split(","))
    for policy_name in policies:
        cfg = getattr(Cf, args.workload)(device_memory_ratio=0.5, max_tokens=args.batch,
                                         policy=Cf.POLICY_LRU if policy_name == "lru" else Cf.POLICY_LFU_INCACHE)
        cfg.num_layers = L
        E, K, H = cfg.num_experts, cfg.top_k, cfg.hidden
        probe = MoEEngine(cfg)
        _, _, tot = probe.expert_layout(0)
        slot_bytes = probe.stats()["slot_bytes"]
        probe.close()
        cfg.device_memory_bytes = int(args.cache_frac * L * E) * slot_bytes
        eng = MoEEngine(cfg)
        es = 2 if eng.dtype == torch.bfloat16 else 4
        g = torch.Generator(device=dev)
        for l in range(L):
            for e in range(E):
                eng.register_expert(l, e, None)
                g.manual_seed(1234 + 1000 * l + e)
                eng.expert_host_view(l, e).view(eng.dtype).copy_(torch.empty(tot // es, dtype=eng.dtype, device=dev).normal_(0, 0.02, generator=g))
            if cfg.shared_inter:
                _, sizs, _ = eng.expert_layout(1)
                eng.register_shared(l, [torch.empty(s // es, dtype=eng.dtype).normal_(0, 0.02) for s in sizs])
        # sequence-specific skew: feature 0..types-1 of an activation is a one-hot of the sequence's type,
        # gate[:, ty] carries that type's Zipf logit bias (a different expert ranking per type and layer)
        rng = np.random.default_rng(7)
        gates = []
        for l in range(L):
            gw = (torch.randn(E, H, generator=torch.Generator().manual_seed(4321 + l)) * 0.02)
            for ty in range(args.types):
                rank = rng.permutation(E) + 1
                gw[:, ty] = torch.from_numpy(-args.zipf * 2.0 * np.log(rank)).float()
            gates.append(gw.to(eng.gate_dtype).to(dev))

        def x_of(seq, step, l):
            x = acts(args.batch, H, eng.dtype, 2024 + l + 1000 * step + 100000 * seq)
            x[:, : args.types] = 0.0
            for i in range(args.batch):
                x[i, (seq + i) % args.types] = 1.0
            return x.to(dev)

        tracer = ExpertTracer(max(args.hist_seqs, 4), L, E)
        use_pf = "prefetch" in policy_name
        engine_pred = "engine_predictor" in policy_name  # the tracer attached to the engine: no read-back, no Python per layer
        lookahead = int(policy_name.split("_la")[1].split("+")[0]) if engine_pred else 0
        if use_pf or engine_pred:
            # history: EAMs of earlier sequences drawn from the same routing distribution (routing only, no FFN)
            hist = np.zeros((args.hist_seqs, L, E), np.float32)
            for s in range(args.hist_seqs):
                for step in range(16):
                    for l in range(L):
                        eng.forward(l, x_of(100 + s, step, l), gates[l], flags=FWD_ROUTE_ONLY)
                        for i in eng.routing()["topk_idx"].reshape(-1):
                            hist[s, l, i] += 1
            tracer.load_trace(hist)
        native = None
        if engine_pred:
            from moe_infinity_amd.engine import ExpertTracerNative

            native = ExpertTracerNative(L, E, max(args.hist_seqs, 4))
            native.load_trace(hist)
        pred = ExpertPredictor(L, E)
        pred.add_tracer(tracer)
        pf = ExpertPrefetcher(L, E, tracer)
        pf.set_archer_engine(eng)
        if policy_name.endswith("+governor"):
            eng.set_prefetch_governor(0.5, 16)
        out = torch.empty(args.batch, H, dtype=eng.dtype, device=dev)
        naive = policy_name.endswith("_all")

        def run_sequence(sid, steps, prefetch):
            if engine_pred and prefetch:
                nseq = native.create_entry()
                eng.set_predictor(native, nseq, lookahead_layers=lookahead, min_share=args.min_share, max_experts=args.max_prefetch)
                for step in range(steps):
                    for l in range(L):
                        for _ in range(reps):
                            a @ b  # attention stand-in on the compute stream
                        eng.forward(l, x_of(sid, step, l), gates[l], out=out)
                eng.set_predictor(None)
                native.finish_entry(nseq)
                return
            seq = tracer.create_entry()
            for step in range(steps):
                for l in range(L):
                    for _ in range(reps):
                        a @ b  # attention stand-in on the compute stream
                    eng.forward(l, x_of(sid, step, l), gates[l], out=out)
                    if prefetch and l + 1 < L:
                        idx = eng.routing()["topk_idx"].reshape(-1)
                        m = pred.predict(seq, idx, l)
                        if naive:
                            pf.prefetch_experts(l + 1, m)
                        else:
                            pf.prefetch_experts(l + 1, m, max_experts=args.max_prefetch, min_share=args.min_share)

        for sid in range(args.types):  # warm-up: one short sequence of every type, on demand
            run_sequence(1000 + sid, 3, False)
        eng.sync_copies()
        torch.cuda.synchronize()
        eng.clear_expert_cache_counts()
        eng.reset_stats()
        t0 = time.perf_counter()
        for sid in range(args.seqs):
            eng.clear_expert_cache_counts()  # prefill->decode boundary of every sequence (interface_example.py:39)
            run_sequence(sid, args.seq_len, use_pf or engine_pred)
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        eng.sync_copies()
        st = eng.stats()
        args.steps = args.seqs * args.seq_len
        res = {"policy": policy_name, "workload": args.workload, "batch": args.batch, "note": args.note, "layers": L, "cache_slots": st["slots_total"], "experts": L * E,
               "sequence_types": args.types, "sequences": args.seqs, "tokens": args.steps,
               "zipf": args.zipf, "attn_standin_us": round(one * reps, 1), "ms_per_token": round(el * 1e3 / args.steps, 3),
               "compute_only_ms_per_token": round(L * one * reps / 1e3, 3),
               "hit_rate": round(st["expert_hits"] / max(1, st["expert_hits"] + st["expert_misses"]), 4),
               "misses": st["expert_misses"], "prefetch_issued": st["prefetch_issued"], "prefetch_useful": st["prefetch_useful"],
               "prefetch_throttled": st["prefetch_throttled"],
               "h2d_GiB": round(st["h2d_bytes"] / 2**30, 2),
               "h2d_GBps_while_busy": round(st["h2d_bytes"] / st["h2d_busy_ms"] / 1e6, 2) if st["h2d_busy_ms"] > 0 else None,
               "copy_busy_ms": round(st["h2d_busy_ms"], 1), "exposed_wait_ms": round(st["exposed_wait_ms"], 1),
               "overlap": round(max(0.0, 1 - st["exposed_wait_ms"] / st["h2d_busy_ms"]), 4) if st["h2d_busy_ms"] > 0 else None}
        print(json.dumps(res), flush=True)
        results.append(res)
        eng.close()


if __name__ == "__main__":
    main()
