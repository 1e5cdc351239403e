"""The parity bars, in one place.  TEST INFRASTRUCTURE ONLY (see moe_ref.py): imported by ``tests/helpers.py``,
``__graft_entry__.smoke()`` and the parity leg of ``bench.py`` — the same bar everywhere.

Bars
  * routing indices / counts / offsets / permutations: bit-exact (checked by the callers with array_equal);
  * per-expert FFN rows: 1 ulp of the model dtype (``rows_report``);
  * block outputs (``block_report``): per element ``|err| <= ulp * (2 * sum_k |contribution_k| + |result|)``
    (bf16 ulp = 2^-7 relative; fp32 2e-5 relative) and mean relative error <= 1e-3 (north_star's tolerance).
    An expert output that differs by one rounding flip (fp32 accumulation order) passes through two more
    roundings in the combine (Tr(y*w), Tr(acc + prod)), each able to move the value by one ulp at ITS OWN scale.

NLLB passthrough discontinuity.  The reference block ends with ``next_states[next_states == 0] = hidden_states[...]``
(moe_infinity/models/nllb_moe.py:103).  In bf16 two weighted expert outputs of opposite sign cancel to EXACTLY 0
for about one element in a few hundred, and the reference then returns the INPUT value there.  A one-ulp flip in
either expert output turns that exact 0 into a tiny non-zero sum (or the reverse), so the two sides legitimately
differ by ``|x|`` — a full-size value — on such an element.  The bar handles this explicitly: where the oracle's
pre-passthrough sum is within the combine tolerance of 0, the result must be EITHER within tolerance of that
pre-passthrough sum OR bit-equal to the input element (the passthrough taken); everywhere else the ordinary bar
applies.  ``block_report`` counts those elements (``passthrough_ambiguous``) so a run can show how many there were.
"""
from __future__ import annotations

import torch


def ulp_of(dtype) -> float:
    return 2e-5 if dtype == torch.float32 else 2.0 ** -7


def block_magnitude(ref) -> torch.Tensor:
    """Per output element: sum of |weighted expert contributions| (+ |shared expert|)."""
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


def block_report(got: torch.Tensor, ref, dtype, golden: torch.Tensor = None, x: torch.Tensor = None) -> dict:
    """Compare a block output with the oracle's (or with ``golden``, the reference block's own output).
    ``x``: the block input, needed only for the NLLB passthrough rule (defaults to ref.extra['x'])."""
    want = (golden if golden is not None else ref.out).float().cpu().reshape(ref.out.shape)
    got = got.float().cpu().reshape(ref.out.shape)
    u = ulp_of(dtype)
    mag = block_magnitude(ref)
    scale = 2.0 * mag + torch.maximum(want.abs(), want.abs().mean())
    tol = scale * u + 1e-30
    err = (got - want).abs()
    ambiguous = 0
    if "pre_passthrough" in ref.extra:
        pre = ref.extra["pre_passthrough"].float().reshape(ref.out.shape)
        xin = (x if x is not None else ref.extra["x"]).float().cpu().reshape(ref.out.shape)
        tol_pre = (2.0 * mag + torch.maximum(pre.abs(), want.abs().mean())) * u + 1e-30
        near_zero = pre.abs() <= tol_pre  # either side may or may not have hit the exact 0
        ok_here = ((got - pre).abs() <= tol_pre) | (got == xin)
        err = torch.where(near_zero, torch.where(ok_here, torch.zeros_like(err), err), err)
        ambiguous = int(near_zero.sum())
    ratio = err / tol
    bad = ratio > 1.0
    denom = want.abs().mean().item() + 1e-30
    rel = err.mean().item() / denom
    worst = float(ratio.max()) if ratio.numel() else 0.0
    # plain figures next to the bar: largest absolute error, and largest error relative to max(|want|, mean|want|)
    # (elements on the NLLB passthrough discontinuity that legitimately took the other branch are excluded, as above)
    floor = torch.maximum(want.abs(), want.abs().mean()) + 1e-30
    rep = {"worst": worst, "n_bad": int(bad.sum()), "n": int(bad.numel()), "mean_rel": rel,
           "max_abs_err": float(err.max()) if err.numel() else 0.0,
           "max_rel_err": float((err / floor).max()) if err.numel() else 0.0,
           "passthrough_ambiguous": ambiguous, "ok": (not bool(bad.any())) and rel <= 1e-3}
    if bad.any():
        i = int(ratio.reshape(-1).argmax())
        rep["worst_at"] = {"flat_index": i, "got": float(got.reshape(-1)[i]), "want": float(want.reshape(-1)[i]),
                           "tol": float(tol.reshape(-1)[i])}
    return rep


def rows_report(got: torch.Tensor, ref: torch.Tensor, dtype, ulps: float = 1.0) -> dict:
    """1 ulp of the model dtype at the element's magnitude (or at the tensor's typical magnitude where terms
    cancel); tight relative bound for fp32."""
    got, ref = got.float().cpu(), ref.float().cpu()
    mag = torch.maximum(torch.maximum(ref.abs(), got.abs()), ref.abs().mean())
    tol = mag * ulp_of(dtype) * ulps + 1e-30
    err = (got - ref).abs()
    bad = err > tol
    rel = err.mean().item() / (ref.abs().mean().item() + 1e-30)
    return {"worst": float((err / tol).max()) if err.numel() else 0.0, "n_bad": int(bad.sum()), "n": int(bad.numel()),
            "mean_rel": rel, "ok": (not bool(bad.any())) and rel <= 1e-3}
