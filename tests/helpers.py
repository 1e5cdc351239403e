"""Shared helpers for the parity tests (test infrastructure)."""
import os

import numpy as np
import torch

from oracle import moe_ref as R
from oracle import parity as P
from oracle.synth import acts, checksum, make_weights

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name), allow_pickle=False)


def tt(a, dtype):
    return torch.from_numpy(np.asarray(a)).to(dtype)


def assert_model_close(got, ref, dtype, what, ulps=1.0):
    """Per-expert FFN rows / routing weights: 1 ulp of the model dtype (oracle/parity.py:rows_report)."""
    rep = P.rows_report(got, ref, dtype, ulps)
    assert rep["n_bad"] == 0, f"{what}: {rep['n_bad']}/{rep['n']} elements beyond {ulps} ulp, worst {rep['worst']:.2f} ulp"
    assert rep["mean_rel"] <= 1e-3, f"{what}: mean relative error {rep['mean_rel']:.2e} > 1e-3"


def engine_for(family, h, f, e, k, dtype, n_shared=0, max_tokens=64, **kw):
    from moe_infinity_amd import MoEEngine
    from moe_infinity_amd import config as Cf

    dt = Cf.DTYPE_BF16 if dtype == torch.bfloat16 else (Cf.DTYPE_F16 if dtype == torch.float16 else Cf.DTYPE_F32)
    et = {"mixtral": Cf.EXPERT_MIXTRAL, "deepseek": Cf.EXPERT_DEEPSEEK, "switch": Cf.EXPERT_SWITCH, "nllb": Cf.EXPERT_NLLB, "fsgpt": Cf.EXPERT_FSGPT,
          "switchgated": Cf.EXPERT_SWITCH_GATED}[family]
    rk = {"mixtral": Cf.ROUTER_MIXTRAL, "deepseek": Cf.ROUTER_DEEPSEEK, "switch": Cf.ROUTER_SWITCH, "nllb": Cf.ROUTER_NLLB, "fsgpt": Cf.ROUTER_NLLB,
          "switchgated": Cf.ROUTER_SWITCH}[family]
    base = dict(num_layers=1, num_experts=e, expert_type=et, hidden=h, inter=f, top_k=k, router_kind=rk, dtype=dt,
                shared_inter=f * n_shared, device_memory_ratio=0.5, max_tokens=max_tokens)
    base.update(kw)
    return MoEEngine(Cf.EngineConfig(**base))


def register_all(eng, experts, shared=None, layer=0):
    for i, ex in enumerate(experts):
        eng.register_expert(layer, i, ex)
    if shared:
        eng.register_shared(layer, shared)


def oracle_expert_rows(ref: R.BlockResult, e_total):
    """Oracle per-expert outputs concatenated in expert-sorted row order."""
    rows = [ref.expert_out[e] for e in range(e_total) if e in ref.expert_out]
    return torch.cat(rows, 0) if rows else None


def assert_block_close(got, ref: R.BlockResult, dtype, what, golden=None):
    """Block-output bar (oracle/parity.py:block_report — the same function bench.py's parity leg calls), incl. the
    explicit rule for NLLB's `== 0` passthrough discontinuity.  `golden`: optionally the reference block's own
    output to compare against instead of the oracle's."""
    rep = P.block_report(got, ref, dtype, golden=golden)
    assert rep["n_bad"] == 0, f"{what}: {rep['n_bad']}/{rep['n']} elements beyond tolerance, worst {rep['worst']:.2f}x at {rep.get('worst_at')}"
    assert rep["mean_rel"] <= 1e-3, f"{what}: mean relative error {rep['mean_rel']:.2e} > 1e-3"
    return rep


# ---- full-size layers: weights generated on the GPU straight into the engine's pinned arena ----------------
_SHAPES = {"mixtral": lambda h, f: [(f, h), (h, f), (f, h)], "deepseek": lambda h, f: [(f, h), (f, h), (h, f)],
           "switch": lambda h, f: [(f, h), (h, f)], "nllb": lambda h, f: [(f, h), (f,), (h, f), (h,)]}


def fill_layer_on_gpu(eng, family, layer, seed, dev="cuda:0", std=0.02):
    """N(0, std^2) expert weights for every expert of `layer`, generated on the GPU and written into the engine's
    pinned host arena (zero-copy registration).  Returns (experts, shared): CPU tensors for the oracle — the
    experts are zero-copy views of the arena in the reference's blob order."""
    cfg = eng.cfg
    off, siz, tot = eng.expert_layout(0)
    dt = eng.dtype
    es = 4 if dt == torch.float32 else 2
    g = torch.Generator(device=dev)
    experts = []
    for e in range(cfg.num_experts):
        eng.register_expert(layer, e, None)
        g.manual_seed(seed + e)
        blob = torch.empty(tot // es, dtype=dt, device=dev).normal_(0.0, std, generator=g)
        raw = eng.expert_host_view(layer, e)
        raw.view(dt).copy_(blob)
        experts.append([raw[o:o + s].view(dt).reshape(sh) for o, s, sh in zip(off, siz, _SHAPES[family](cfg.hidden, cfg.inter))])
    shared = None
    if cfg.shared_inter:
        _, sizs, _ = eng.expert_layout(1)
        g.manual_seed(seed + 9999)
        shared = [torch.empty(s // es, dtype=dt, device=dev).normal_(0.0, std, generator=g).cpu().reshape(sh)
                  for s, sh in zip(sizs, _SHAPES[family](cfg.hidden, cfg.shared_inter))]
        eng.register_shared(layer, shared)
    torch.cuda.synchronize()
    return experts, shared


def assert_as_accurate_as_the_oracle(got, ref: R.BlockResult, family, x3d, experts, dtype, what, shared=None, rows=None):
    """The fp32-exact arm (oracle/parity.py): the GPU block output — and, with ``rows`` = the GPU's per-expert FFN rows in
    expert-sorted order, those rows too — must be as close to the fp32 computation as the reference's CPU path (the oracle in
    the model dtype) is: mean |gpu - exact| <= 1.15 x mean |oracle - exact| (fp32 models: 4 x, oracle/parity.py exact_arm_factor)."""
    ex = P.exact_block(family, x3d, ref, experts, shared=shared)
    rep = P.accuracy_report(got, ref, ex, dtype)
    assert rep["ok"], (f"{what}: the GPU result is further from the fp32-exact block ({rep['gpu_vs_exact']:.3e}) than the oracle in the model dtype "
                       f"({rep['oracle_vs_exact']:.3e}), ratio {rep['ratio']:.3f} > {rep['factor']}")
    if rows is not None:
        order = [e for e in sorted(ref.expert_out)]
        rr = P.rows_accuracy_report(rows, torch.cat([ref.expert_out[e] for e in order], 0), torch.cat([ex["rows"][e] for e in order], 0))
        assert rr["ok"], f"{what}: expert FFN rows further from fp32-exact ({rr['gpu_vs_exact']:.3e}) than the oracle's ({rr['oracle_vs_exact']:.3e})"
        rep["rows"] = rr
    return rep
