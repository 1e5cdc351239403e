"""Seeded synthetic weights/activations shared by the golden generator, the parity tests
and bench.py (SURVEY.md section 8d protocol).  TEST INFRASTRUCTURE ONLY (see moe_ref.py).

Golden fixtures store only seeds + a checksum of the weights; tests regenerate the
weights with these helpers and verify the checksum before trusting the fixture."""
import numpy as np
import torch


def randw(shape, dtype, gen, std=0.02):
    return (torch.randn(*shape, generator=gen) * std).to(dtype)


def acts(t, h, dtype, seed):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(t, h, generator=g)
    x = x / x.pow(2).mean(-1, keepdim=True).sqrt()
    return x.to(dtype)


def make_weights(family, h, f, e, seed, dtype, n_shared=0, gate_std=0.02):
    """Returns (gate[E,H], experts: list of tensor lists in the reference's blob order,
    shared or None).  Generation order is part of the fixture contract - do not reorder."""
    g = torch.Generator().manual_seed(seed)
    gate = randw((e, h), dtype, g, std=gate_std)
    experts, shared = [], None
    if family == "mixtral":
        for _ in range(e):
            w1 = randw((f, h), dtype, g); w2 = randw((h, f), dtype, g); w3 = randw((f, h), dtype, g)
            experts.append([w1, w2, w3])
    elif family == "deepseek":
        def mlp(fi):
            return [randw((fi, h), dtype, g), randw((fi, h), dtype, g), randw((h, fi), dtype, g)]
        for _ in range(e):
            experts.append(mlp(f))
        if n_shared:
            shared = mlp(f * n_shared)
    elif family == "switch":
        for _ in range(e):
            experts.append([randw((f, h), dtype, g), randw((h, f), dtype, g)])
    elif family == "nllb":
        for _ in range(e):
            experts.append([randw((f, h), dtype, g), randw((f,), dtype, g), randw((h, f), dtype, g), randw((h,), dtype, g)])
    else:
        raise ValueError(family)
    return gate, experts, shared


def checksum(gate, experts, shared=None):
    tot = [gate.double().sum().item(), gate.double().pow(2).sum().item()]
    for ex in experts + ([shared] if shared else []):
        for t in ex:
            tot[0] += t.double().sum().item()
            tot[1] += t.double().pow(2).sum().item()
    return np.array(tot, dtype=np.float64)
