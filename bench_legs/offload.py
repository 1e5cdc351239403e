"""bench.py's OFFLOAD-REGIME leg (BASELINE configs 2 and 3), moved out of bench.py's run_workload in round 6: the same engine
with the expert cache cut to a byte budget — on-demand fetches, the routing x replacement-policy matrix, the measured attention
time between the MoE layers, and speculation three ways (none | the engine-side EAM predictor | the next-layer gate lookahead) on
independent and on residual-stream activations.  `offload_regime(c)` takes the run's state as a namespace (the locals of
run_workload it reads are listed in NEEDS) and returns the `miss_heavy` / `offload_regime` object of the details file."""
import time

import torch

HBM_PEAK_GBS = 8000.0
PCIE_GBS = 63.0
NEEDS = ('offload_frac', 'B', 'Cf', 'E', 'H', 'K', 'L', 'args', 'batch_rows', 'dev', 'dt', 'eng', 'family', 'gates', 'label', 'main', 'ms_per_step', 'nsteps', 'out', 'rank', 'st', 'steps', 'use_ep', 'warmup', 'world', 'xs')


def offload_regime(c):
    offload_frac, B, Cf, E, H, K, L, args, batch_rows, dev, dt, eng, family, gates, label, main, ms_per_step, nsteps, out, rank, st, steps, use_ep, warmup, world, xs = c.offload_frac, c.B, c.Cf, c.E, c.H, c.K, c.L, c.args, c.batch_rows, c.dev, c.dt, c.eng, c.family, c.gates, c.label, c.main, c.ms_per_step, c.nsteps, c.out, c.rank, c.st, c.steps, c.use_ep, c.warmup, c.world, c.xs
    # ---- offload regime (BASELINE configs 2 and 3): the same engine with the expert cache cut to a byte budget.
    # One sub-leg = (routing, replacement policy, attention stand-in, speculation): cache flushed, two settling steps, counters
    # reset, `msteps` decode steps timed.  Routing "natural" = what the random gate produces (uniform over experts: the hit
    # rate is the capacity fraction whatever the policy); "zipf1.2" = SURVEY.md section 8d's skew: every layer ranks its experts
    # (fixed seed) and adds -1.2 ln(rank) to their logits (coordinate 0 of every activation is a constant 4, the gate's column
    # 0 carries the bias / 4), so a few experts per layer are hot — where LFU-in-cache and LRU can differ.
    miss = None
    if rank == 0 and world == 1 and not use_ep and offload_frac > 0 and not args.budget_gib:
        import numpy as np

        slot = st["slot_bytes"]
        budget = int(offload_frac * L * E * slot)
        msteps = max(5, min(steps, args.miss_heavy_steps))
        rng = np.random.default_rng(7)
        gates_z = []
        for l in range(L):
            gz = gates[l].clone()
            gz[:, 0] = torch.from_numpy(-1.2 * np.log(rng.permutation(E) + 1.0) / 4.0).to(gz.dtype).to(dev)
            gates_z.append(gz)

        def x_zipf(s, l):
            x = xs[warmup + (s % steps)][l].clone()
            x[:, 0] = 4.0
            return x

        # attention stand-in (the MoE layers of a real model sit between attention blocks: profiles/r04_attention_block_time_*,
        # tools/attn_time.py): a bf16 matmul on the compute stream, repeated to the measured time
        attn_us = args.offload_attn_us if family == "deepseek" else 0.0  # (the measured attention time is DeepSeek-V2-Lite's)
        reps, one_us = 0, 0.0
        if attn_us > 0:
            ma = torch.randn(2048, 2048, device=dev, dtype=torch.bfloat16)
            mb = torch.randn(2048, 2048, device=dev, dtype=torch.bfloat16)
            for _ in range(5):
                ma @ mb
            torch.cuda.synchronize(dev)
            t0 = time.perf_counter()
            for _ in range(50):
                ma @ mb
            torch.cuda.synchronize(dev)
            one_us = (time.perf_counter() - t0) / 50 * 1e6
            reps = max(1, int(round(attn_us / one_us)))

        # speculation legs: a synthetic RESIDUAL stream.  The default inputs are independent per layer (seed 2024 + layer), so
        # nothing a layer sees says anything about the next one; a transformer's hidden state changes little from one MoE layer
        # to the next.  x_0 = the default rows of layer 0, x_{l+1} = rmsnorm(x_l + RES_EPS * n_l), n_l ~ N(0, 1) seeded:
        # cos(x_l, x_{l+1}) = 1 / sqrt(1 + RES_EPS^2) = 0.894 at RES_EPS = 0.5.
        RES_EPS = 0.5
        xres = {}

        def x_res(s, l):
            key = (s % steps, l)
            if key not in xres:
                if l == 0:
                    xres[key] = xs[warmup + (s % steps)][0].float()
                else:
                    gg = torch.Generator().manual_seed(777000 + 1000 * (s % steps) + l)
                    x = x_res(s, l - 1) + RES_EPS * torch.randn(B, H, generator=gg).to(dev)
                    xres[key] = x / x.pow(2).mean(-1, keepdim=True).sqrt()
            return xres[key]

        def offload_leg(routing, policy, with_attn, speculate, nsteps_leg, stream="independent", la_max=0):
            zipf = routing != "natural"
            gl = gates_z if zipf else gates
            residual = stream == "residual"

            def x_of(s, l):
                if residual:
                    x = x_res(s, l).to(dt)
                    if zipf:
                        x = x.clone()
                        x[:, 0] = 4.0
                    return x
                return x_zipf(s, l) if zipf else xs[warmup + (s % steps)][l]
            eng.set_cache_policy(Cf.POLICY_LRU if policy == "lru" else Cf.POLICY_LFU_INCACHE)
            eng.set_cache_budget(slot)      # flush: one slot ...
            eng.set_cache_budget(budget)    # ... and back to the budget of this leg
            native, nseq = None, -1
            if speculate == "lookahead":
                # next-layer gate lookahead (moeinf_set_lookahead): layer l+1's gate over layer l's input rows, the predicted
                # experts issued behind layer l's misses on the same copy stream
                eng.set_lookahead(gl, max_experts=la_max or 2 * K)
            elif speculate:
                # the engine-side predictor (moeinf_set_predictor; reference: memory/expert_tracer.py + expert_predictor.py +
                # expert_prefetcher.py): history = activation matrices of 8 earlier sequences under the same routing
                from moe_infinity_amd.engine import FWD_ROUTE_ONLY, ExpertTracerNative

                hist = np.zeros((8, L, E), np.float32)
                for sq in range(8):
                    for s in range(8):
                        for l in range(L):
                            eng.forward(l, x_of(sq * 8 + s, l) if (zipf or residual) else xs[(sq * 8 + s) % nsteps][l], gl[l], batch_rows=batch_rows, flags=FWD_ROUTE_ONLY)
                            for i in eng.routing()["topk_idx"].reshape(-1):
                                if i >= 0:
                                    hist[sq, l, i] += 1
                native = ExpertTracerNative(L, E, 8)
                native.load_trace(hist)
                nseq = native.create_entry()
                # (min_share 0.05: with K of E = 6 of 64 an expert that EVERY token picks has 1/6 of its layer's activations; the
                # reference's prefetcher has no threshold at all — it enqueues every predicted expert)
                eng.set_predictor(native, nseq, lookahead_layers=2, min_share=0.05, max_experts=16)
                eng.set_prefetch_governor(0.5, 16)

            def steps_(s0, n):
                for s in range(s0, s0 + n):
                    for l in range(L):
                        if with_attn:
                            for _ in range(reps):
                                ma @ mb
                        eng.forward(l, x_of(s, l), gl[l], batch_rows=batch_rows, out=out)

            steps_(0, 2)  # settle the cache
            eng.sync_copies()
            torch.cuda.synchronize(dev)
            eng.clear_expert_cache_counts()
            eng.reset_stats()
            t0 = time.perf_counter()
            steps_(2, nsteps_leg)
            torch.cuda.synchronize(dev)
            el = time.perf_counter() - t0
            eng.sync_copies()
            s_ = eng.stats()
            if speculate == "lookahead":
                eng.set_lookahead(None)
            elif speculate:
                eng.set_predictor(None)
                native.finish_entry(nseq)
                eng.set_prefetch_governor(0.0, 16)
            mis = s_["expert_misses"]
            link = s_["h2d_bytes"] / s_["h2d_busy_ms"] / 1e6 if s_["h2d_busy_ms"] > 0 else None
            attn_ms = L * reps * one_us / 1e3 if with_attn else 0.0
            spec_name = {False: "none (on-demand fetches only)", True: "engine predictor, lookahead 2, min_share 0.05, governor 0.5",
                         "lookahead": f"next-layer gate lookahead, the {la_max or 2 * K} most confident predictions per layer, issued behind the layer's misses"}[speculate]
            done = s_["prefetch_useful"] + s_.get("prefetch_wasted", 0)
            return {"routing": routing, "policy": policy, "speculation": spec_name,
                    "speculation_kind": {False: "none", True: "eam", "lookahead": f"gate-lookahead top{la_max or 2 * K}"}[speculate],
                    "activations": f"residual stream x_(l+1) = rmsnorm(x_l + {RES_EPS} n_l), cos 0.894" if residual else "independent per layer (seed 2024 + layer)",
                    "prefetch_precision": None if not s_["prefetch_issued"] else round(min(1.0, s_["prefetch_useful"] / max(1, s_["prefetch_issued"])), 4),  # (copies issued in the settling steps can be used after the counters were reset: capped at 1)
                    "prefetch_wasted": s_.get("prefetch_wasted"), "prefetch_settled": done,
                    "attention_standin_us_per_layer": round(reps * one_us, 1) if with_attn else 0.0,
                    "steps": nsteps_leg, "ms_per_token": round(el * 1e3 / nsteps_leg / B, 3),
                    "moe_ms_per_token_without_the_standin": round(el * 1e3 / nsteps_leg / B - attn_ms / B, 3),
                    "hit_rate": round(s_["expert_hits"] / max(1, s_["expert_hits"] + mis), 4), "misses_per_token": round(mis / nsteps_leg / B, 2),
                    "prefetch_issued": s_["prefetch_issued"], "prefetch_useful": s_["prefetch_useful"],
                    "h2d_GiB": round(s_["h2d_bytes"] / 2**30, 2), "h2d_link_busy_ms": round(s_["h2d_busy_ms"], 1),
                    "h2d_GBps": None if link is None else round(link, 2),
                    "h2d_frac_of_pcie5_x16": None if link is None else round(link / PCIE_GBS, 3),
                    "h2d_frac_of_hbm_peak": None if link is None else round(link / HBM_PEAK_GBS, 4),
                    "exposed_wait_ms": round(s_["exposed_wait_ms"], 1),
                    "overlap": None if s_["h2d_busy_ms"] <= 0 else round(max(0.0, 1.0 - s_["exposed_wait_ms"] / s_["h2d_busy_ms"]), 4),
                    "_raw": (el, s_, mis, link)}

        eng.set_cache_budget(budget)
        base = offload_leg("natural", args.policy, False, False, msteps)
        mel, ms_, misses, link = base.pop("_raw")
        link_bytes = ms_["h2d_bytes"] / max(1, misses)  # bytes one miss moves over the link (= the host blob: half a slot for fp8 experts)
        bound_ms = misses * link_bytes / (56.0e9) * 1e3 / msteps  # every miss crosses the link once at the measured 56 GB/s
        miss = {"what": f"{label}, expert cache = {offload_frac:.0%} of the expert bytes ({budget / 2**30:.1f} GiB, "
                        f"{ms_['slots_total']} of {L * E} experts), on-demand fetches only, natural routing, {args.policy}",
                **{k: v for k, v in base.items() if k not in ("routing", "policy", "speculation", "attention_standin_us_per_layer", "moe_ms_per_token_without_the_standin")},
                "tokens_per_s": round(B * msteps / mel, 3),
                "pcie_bound_ms_per_token": round(bound_ms / B, 3),
                "ms_per_token_over_pcie_bound": round(mel * 1e3 / msteps / max(bound_ms, 1e-9), 3),
                # why `overlap` is what it is: one miss is `copy_ms_per_miss` of link time, the compute stream has
                # `compute_ms_per_layer` of MoE work per layer to put beside it (measured above, every expert cached), and on
                # demand the copy can only start once the layer has routed — the layer waits for the rest of it
                "copy_ms_per_miss": None if link is None else round(link_bytes / (link * 1e9) * 1e3, 3),
                "compute_ms_per_layer": round(ms_per_step / L, 4),
                "overlap_ceiling_on_demand": None if link is None else round(min(1.0, (ms_per_step / L) / max(1e-9, (misses / msteps / L) * link_bytes / (link * 1e9) * 1e3)), 4),
                "physics": "on-demand: a miss is issued when its layer routes and the layer's FFN needs it at once, so at most "
                           "compute_ms_per_layer of every (misses_per_layer x copy_ms_per_miss) can overlap (overlap_ceiling_on_demand); "
                           "hiding more needs copies issued LAYERS ahead (speculation: the sub-legs below / profiles/r04_prefetch_study_*)"}
        # the matrix SURVEY.md section 8d asks for: routing x replacement policy, same engine, same budget, same steps
        zsteps = max(msteps, args.offload_zipf_steps)
        legs = []
        for routing, policy in (("natural", "lru"), ("zipf1.2", "lfu_incache"), ("zipf1.2", "lru")):
            lg = offload_leg(routing, policy, False, False, zsteps if routing != "natural" else msteps)
            lg.pop("_raw")
            legs.append(lg)
        first = dict(base)
        legs.insert(0, first)
        if attn_us > 0:  # BASELINE config 2: "prefetch stream overlap" with real attention time between the MoE layers
            for speculate in (False, True):
                lg = offload_leg("zipf1.2", "lfu_incache", True, speculate, zsteps)
                lg.pop("_raw")
                legs.append(lg)
            # ... and on a residual stream, where the next layer IS predictable: on demand / history (EAM) / next-layer gate
            for routing in ("natural", "zipf1.2"):
                for speculate, la_max in ((False, 0), (True, 0), ("lookahead", 1), ("lookahead", 2), ("lookahead", K)):
                    lg = offload_leg(routing, "lfu_incache", True, speculate, zsteps if routing != "natural" else msteps, stream="residual", la_max=la_max)
                    lg.pop("_raw")
                    legs.append(lg)
        miss["sub_legs"] = legs
        eng.set_cache_policy(Cf.POLICY_LRU if args.policy == "lru" else Cf.POLICY_LFU_INCACHE)

    return miss
