"""Shared helpers for the parity tests (test infrastructure)."""
import os

import numpy as np
import torch

from oracle import moe_ref as R
from oracle.synth import acts, checksum, make_weights

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name), allow_pickle=False)


def tt(a, dtype):
    return torch.from_numpy(np.asarray(a)).to(dtype)


def ulp_tol(ref: torch.Tensor, got: torch.Tensor, dtype) -> torch.Tensor:
    """Elementwise tolerance: 1 ulp of the model dtype at the element's magnitude, or at the
    tensor's typical magnitude where terms cancel (bf16), tight relative bound for fp32."""
    ref, got = ref.float(), got.float()
    mag = torch.maximum(torch.maximum(ref.abs(), got.abs()), ref.abs().mean())
    if dtype == torch.float32:
        return mag * 2e-5 + 1e-30
    return mag * 2.0 ** -7 + 1e-30


def assert_model_close(got, ref, dtype, what, ulps=1.0):
    """ulps=1 for a single rounded op chain (expert FFN rows, routing weights); block outputs sum K
    rounded contributions, each of which may carry a 1-ulp flip from accumulation order -> ulps=2."""
    got, ref = got.float().cpu(), ref.float().cpu()
    err = (got - ref).abs()
    tol = ulp_tol(ref, got, dtype) * ulps
    bad = err > tol
    assert not bool(bad.any()), f"{what}: {int(bad.sum())}/{bad.numel()} elements beyond 1 ulp, worst {float((err / tol).max()):.2f} ulp"
    # bulk agreement: well inside the north-star's 1e-3
    rel = err.mean().item() / (ref.abs().mean().item() + 1e-30)
    assert rel <= 1e-3, f"{what}: mean relative error {rel:.2e} > 1e-3"


def engine_for(family, h, f, e, k, dtype, n_shared=0, max_tokens=64, **kw):
    from moe_infinity_amd import MoEEngine
    from moe_infinity_amd import config as Cf

    dt = Cf.DTYPE_BF16 if dtype == torch.bfloat16 else Cf.DTYPE_F32
    et = {"mixtral": Cf.EXPERT_MIXTRAL, "deepseek": Cf.EXPERT_DEEPSEEK, "switch": Cf.EXPERT_SWITCH, "nllb": Cf.EXPERT_NLLB}[family]
    rk = {"mixtral": Cf.ROUTER_MIXTRAL, "deepseek": Cf.ROUTER_DEEPSEEK, "switch": Cf.ROUTER_SWITCH, "nllb": Cf.ROUTER_NLLB}[family]
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


def block_magnitude(ref: R.BlockResult) -> torch.Tensor:
    """Per output element: sum of |weighted expert contributions| (+ |shared expert|).  One rounding flip
    in a contribution moves the block output by up to one ulp AT THE SCALE OF THAT CONTRIBUTION, which
    can exceed an ulp of the (possibly cancelling) sum."""
    out = ref.out.reshape(-1, ref.out.shape[-1]).float()
    mag = torch.zeros_like(out)
    if ref.weights_mask is not None:
        wm = ref.weights_mask.reshape(out.shape[0], -1).float()
        rm = ref.router_mask.reshape(out.shape[0], -1).bool()
        for e, y in ref.expert_out.items():
            tok = rm[:, e]
            mag[tok] += (y.float() * wm[tok, e][:, None]).abs()
    if "shared_out" in ref.extra:
        mag += ref.extra["shared_out"].reshape(out.shape).float().abs()
    return mag.reshape(ref.out.shape)


def assert_block_close(got, ref: R.BlockResult, dtype, what, golden=None):
    """Block-output bar (bf16): |err| <= ulp * (2 * sum_k |contribution_k| + |result|) per element, ulp =
    2^-7 relative: an expert output that differs by one rounding flip (accumulation order) passes
    through two more roundings (Tr(y*w), Tr(acc + prod)), each able to move the value by one ulp at
    its own scale; plus mean relative error <= 1e-3 (north_star's tolerance).  fp32: 2e-5 relative.
    `golden`: optionally the reference block's own output to compare against instead of the oracle's."""
    want = (golden if golden is not None else ref.out).float().cpu().reshape(ref.out.shape)
    got = got.float().cpu().reshape(ref.out.shape)
    scale = 2.0 * block_magnitude(ref) + torch.maximum(want.abs(), want.abs().mean())
    tol = scale * (2e-5 if dtype == torch.float32 else 2.0 ** -7) + 1e-30
    err = (got - want).abs()
    bad = err > tol
    assert not bool(bad.any()), f"{what}: {int(bad.sum())}/{bad.numel()} elements beyond tolerance, worst {float((err / tol).max()):.2f}x"
    rel = err.mean().item() / (want.abs().mean().item() + 1e-30)
    assert rel <= 1e-3, f"{what}: mean relative error {rel:.2e} > 1e-3"
