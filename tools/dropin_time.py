#!/usr/bin/env python3
"""Time the DROP-IN path: one decode token through L' MoE layers exactly as the reference drives its pybind module —

    router in Python (moe_infinity/models/mixtral.py:42-65, deepseek.py:55-91: ATen ops on the GPU)
    DistributedExpertExecutor.dispatch_local (moe_infinity/distributed/expert_executor.py:32-58):
        router_mask.sum(0).cpu() -> set_inputs -> set_expected_queue -> enqueue_expert x U -> wait_expert
    combine in Python (mixtral.py:96-101, deepseek.py:123-136; DeepSeek adds its shared experts as a dense module)

over ``prefetch_op.prefetch_handle`` / ``prefetch_op.expert_dispatcher`` (the offload directory, tensor ids, topology,
register_expert: what OffloadEngine does at start-up), beside the FUSED path (one moeinf_moe_forward per layer) on the very
same engine and weights.  Reports ms per token, host microseconds per boundary call, and the ratio.

    python tools/dropin_time.py [--workload mixtral-8x7b|deepseek-v2-lite] [--layers 8] [--steps 20] [--dir /tmp/...]

Full layer sizes; ``--layers`` of the model's L layers (the offload directory holds every expert: 8 Mixtral layers = 21 GiB),
per-token figures scaled to the model's L and said so.  Used by bench.py's ``dropin`` leg (measure())."""
import argparse
import json
import os
import shutil
import sys
import tempfile
import time

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

SHAPES = {
    # name: (family, L, E, K, H, F, shared F, expert_type id, tensor order of one expert)
    "mixtral-8x7b": ("mixtral", 32, 8, 2, 4096, 14336, 0, 4),
    "deepseek-v2-lite": ("deepseek", 26, 64, 6, 2048, 1408, 2816, 5),
}


class _Timer:
    def __init__(self):
        self.t = {}
        self.n = {}

    def add(self, name, dt):
        self.t[name] = self.t.get(name, 0.0) + dt
        self.n[name] = self.n.get(name, 0) + 1


class _TimedDispatcher:
    """the pybind object as dispatch_local sees it, every call timed on the host"""

    def __init__(self, disp, timer):
        self._d, self._t = disp, timer

    def __getattr__(self, name):
        fn = getattr(self._d, name)

        def call(*a, **kw):
            t0 = time.perf_counter()
            r = fn(*a, **kw)
            self._t.add(name, time.perf_counter() - t0)
            return r

        return call


def measure(workload="mixtral-8x7b", layers=8, steps=20, warmup=3, directory=None, check_layers=1, log=lambda *a: None):
    from moe_infinity_amd import prefetch_op as P
    from moe_infinity_amd.expert_executor import DistributedExpertExecutor

    family, L_model, E, K, H, Fd, Fs, etype = SHAPES[workload]
    L = min(layers, L_model)
    dev = torch.device("cuda", torch.cuda.current_device())
    dt = torch.bfloat16
    own_dir = directory is None
    directory = directory or tempfile.mkdtemp(prefix="moeinf_dropin_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
    P.configure(max_tokens=8, top_k=K, device_memory_bytes=0, devices=[dev.index])
    handle = P.prefetch_handle(directory, 0.75)
    try:
        gen = torch.Generator(device=dev).manual_seed(1234)
        shapes = [(Fd, H), (H, Fd), (Fd, H)] if family == "mixtral" else [(Fd, H), (Fd, H), (H, Fd)]
        names = ["w1", "w2", "w3"] if family == "mixtral" else ["gate_proj", "up_proj", "down_proj"]
        topo, ex_ids, gates, shared = [], [], [], []
        tid = 0
        t0 = time.time()
        for l in range(L):
            gates.append(torch.empty(E, H, dtype=dt, device=dev).normal_(0.0, 0.02, generator=gen))  # (the gate is a parameter in the model dtype; DeepSeek's MoEGate up-casts it)
            if Fs:
                shared.append([torch.empty(s, dtype=dt, device=dev).normal_(0.0, 0.02, generator=gen) for s in [(Fs, H), (Fs, H), (H, Fs)]])
            ids_l = []
            for e in range(E):
                ids = []
                for nm, shp in zip(names, shapes):
                    handle.offload(torch.empty(shp, dtype=dt, device=dev).normal_(0.0, 0.02, generator=gen).cpu(), tid)
                    ids.append(tid)
                    tid += 1
                ids_l.append(ids)
            ex_ids.append(ids_l)
            topo.append((f"layers.{l}.moe.experts", ids_l))
        params = {}
        for l in range(L):
            for ids in ex_ids[l]:
                for i in ids:
                    params[i] = torch.nn.Parameter(torch.zeros(1, dtype=dt), requires_grad=False)
                    handle.register(params[i].data, i)
        handle.set_topology(topo)
        disp = P.expert_dispatcher(E, L, 0, etype, 8)
        for l in range(L):
            for e in range(E):
                disp.register_expert(l, e, ex_ids[l][e])
        log(f"offload directory + registration of {L} x {E} experts: {time.time() - t0:.0f}s ({directory})")
        timer = _Timer()
        ex = DistributedExpertExecutor(None)
        ex.set_expert_dispatcher(_TimedDispatcher(disp, timer))
        eng = handle.engine

        def x_of(s, l):
            g = torch.Generator().manual_seed(2024 + l + 1000 * s)
            x = torch.randn(1, H, generator=g)
            return (x / x.pow(2).mean(-1, keepdim=True).sqrt()).to(dt).to(dev)

        xs = [[x_of(s, l) for l in range(L)] for s in range(warmup + steps)]

        def shared_mlp(l, x):  # DeepseekV2MLP (a dense module of the reference: plain ATen)
            g_, u_, d_ = shared[l]
            return F.linear(F.silu(F.linear(x, g_)) * F.linear(x, u_), d_)

        def block_dropin(l, hidden):
            """the reference block's forward, op for op"""
            t0 = time.perf_counter()
            if family == "mixtral":  # mixtral.py:46-65
                router_logits = F.linear(hidden, gates[l])
                rw = F.softmax(router_logits, dim=1, dtype=torch.float)
                rw, sel = torch.topk(rw, K, dim=-1)
                rw /= rw.sum(dim=-1, keepdim=True)
                rw = rw.to(hidden.dtype)
                router_mask = F.one_hot(sel, num_classes=E)
                wmask = (rw[:, :, None] * router_mask).permute(0, 2, 1)
                router_mask = router_mask.permute(0, 2, 1)
                router_mask = torch.logical_or(router_mask[:, :, 0], router_mask[:, :, 1])
                wmask = torch.sum(wmask, dim=-1)
            else:  # MoEGate.forward (modeling_deepseek.py:463-512, greedy, norm_topk_prob False) + deepseek.py:72-91
                logits = F.linear(hidden.type(torch.float32), gates[l].type(torch.float32), None)
                scores = logits.softmax(dim=-1, dtype=torch.float32)
                tw, sel = torch.topk(scores, k=K, dim=-1, sorted=False)
                tw = tw * 1.0
                router_mask = F.one_hot(sel, num_classes=E)
                wmask = (tw[:, :, None] * router_mask).permute(0, 2, 1)
                wmask = torch.sum(wmask, dim=-1)
                router_mask = router_mask.permute(0, 2, 1)
                for i in range(K):
                    router_mask[:, :, 0] = torch.logical_or(router_mask[:, :, 0], router_mask[:, :, i])
                router_mask = router_mask[:, :, 0]
            timer.add("python_router_and_masks", time.perf_counter() - t0)
            final = torch.zeros((hidden.shape[0], H), dtype=hidden.dtype, device=hidden.device)
            t0 = time.perf_counter()
            results = ex.dispatch_local(hidden, router_mask, l)
            timer.add("dispatch_local_total", time.perf_counter() - t0)
            t0 = time.perf_counter()
            for output, _, idx, _ in results:  # mixtral.py:96-101 / deepseek.py:123-129
                tok = router_mask[:, idx].bool()
                final[tok, :] += output.to(wmask.device) * wmask[tok, idx][:, None].to(output.dtype)
            if Fs:
                final = final + shared_mlp(l, hidden)
            timer.add("python_combine", time.perf_counter() - t0)
            return final

        out = torch.empty(1, H, dtype=dt, device=dev)

        def block_fused(l, hidden):
            y = eng.forward(l, hidden, gates[l], out=out)
            return y + shared_mlp(l, hidden) if Fs else y

        def run(block, s0, n):
            for s in range(s0, s0 + n):
                for l in range(L):
                    block(l, xs[s][l])

        res = {}
        for name, block in (("dropin", block_dropin), ("fused", block_fused)):
            run(block, 0, warmup)
            torch.cuda.synchronize(dev)
            timer.t.clear(), timer.n.clear()
            eng.reset_stats()
            t0 = time.perf_counter()
            run(block, warmup, steps)
            torch.cuda.synchronize(dev)
            res[name] = (time.perf_counter() - t0) / steps
            if name == "dropin":
                calls = {k: {"calls_per_layer": round(timer.n[k] / steps / L, 2), "host_us_per_call": round(timer.t[k] / timer.n[k] * 1e6, 1),
                             "host_us_per_layer": round(timer.t[k] / steps / L * 1e6, 1)} for k in sorted(timer.t)}
        # the two product paths against each other (the drop-in path's parity against the ORACLE is tests/test_gpu_dropin.py's
        # job; this tool is not test infrastructure and does not import it): same experts, same routing arithmetic up to the
        # Python router's ATen summation order, so the block outputs agree to a few ulps of the model dtype
        agree = None
        if check_layers:
            agree = True
            for l in range(min(check_layers, L)):
                x = xs[0][l]
                a_, b_ = block_dropin(l, x).float(), block_fused(l, x).float()
                tol = 2.0 ** -6 * torch.maximum(b_.abs(), b_.abs().mean())
                agree = agree and bool(((a_ - b_).abs() <= tol).all())
        parity_ok = agree
        st = eng.stats()
        boundary = [k for k in calls if not k.startswith("python_") and k != "dispatch_local_total"]
        return {
            "what": f"{workload}: one decode token through {L} full-size MoE layers via prefetch_op.expert_dispatcher exactly as "
                    "dispatch_local drives it (Python router + masks, .cpu() sync, set_inputs, set_expected_queue, enqueue_expert x U, "
                    "wait_expert, Python combine), beside the fused path (moeinf_moe_forward) on the same engine and weights",
            "layers_timed": L, "layers_model": L_model, "steps": steps,
            "dropin_ms_per_layer": round(res["dropin"] * 1e3 / L, 4), "fused_ms_per_layer": round(res["fused"] * 1e3 / L, 4),
            "ms_per_token": round(res["dropin"] * 1e3 / L * L_model, 3), "fused_ms_per_token": round(res["fused"] * 1e3 / L * L_model, 3),
            "scaled": None if L == L_model else f"per-layer time x {L_model} layers (timed: {L})",
            "over_fused": round(res["dropin"] / res["fused"], 3),
            "calls_per_token": round(sum(calls[k]["calls_per_layer"] for k in boundary) * L_model, 1),
            "host_us_per_call": round(sum(calls[k]["host_us_per_layer"] for k in boundary) / max(1e-9, sum(calls[k]["calls_per_layer"] for k in boundary)), 1),
            # who spends the drop-in path's host time: this repo's side of the boundary vs the reference's own Python around it
            "boundary_us_per_layer": round(sum(calls[k]["host_us_per_layer"] for k in boundary), 1),
            "reference_python_us_per_layer": round(calls["python_router_and_masks"]["host_us_per_layer"] + calls["python_combine"]["host_us_per_layer"]
                                                   + calls["dispatch_local_total"]["host_us_per_layer"] - sum(calls[k]["host_us_per_layer"] for k in boundary), 1),
            "host_calls": calls, "hit_rate": round(st["expert_hits"] / max(1, st["expert_hits"] + st["expert_misses"]), 4),
            "parity_ok": parity_ok, "parity_is": "drop-in block output == fused block output within 2^-6 relative (first layer)",
        }
    finally:
        handle.clean_up_resources()
        P.configure(devices=None, top_k=0, max_tokens=256)
        if own_dir:
            shutil.rmtree(directory, ignore_errors=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="mixtral-8x7b", choices=sorted(SHAPES))
    ap.add_argument("--layers", type=int, default=8)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--dir", default=None)
    a = ap.parse_args()
    import __graft_entry__ as entry

    entry.build()
    r = measure(a.workload, a.layers, a.steps, directory=a.dir, log=lambda *m: print("[dropin]", *m, file=sys.stderr, flush=True))
    print(json.dumps(r, indent=1))


if __name__ == "__main__":
    main()
