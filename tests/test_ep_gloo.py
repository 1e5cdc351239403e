"""World-size-2 and -4 tests of the expert-parallel host path (moe-infinity_amd/ep.py) on CPU with gloo:
each rank routes its own tokens, rows cross ranks with all_to_all, the result must equal the
single-process oracle block.  Compute steps are supplied by the oracle (tests/ep_cpu_ops.py)."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q, var_threshold=64, tight_capacity=False, t=6):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from ep_cpu_ops import OracleEpOps
        from moe_infinity_amd.ep import ExpertParallelMoE
        from oracle import moe_ref as R
        from oracle.synth import acts, make_weights

        h, f, e, k, L = 128, 256, 8, 2, 2
        ws = [make_weights("mixtral", h, f, e, 50 + l, torch.bfloat16) for l in range(L)]
        ops = OracleEpOps([w[1] for w in ws], rank, world, k, e, h)
        # tight_capacity: tokens * min(K, ceil(E / world)) row slots per peer (world 8 with E = 8: ONE slot per token)
        ep = ExpertParallelMoE(ops, h, k, t, torch.bfloat16, "cpu", var_threshold=var_threshold, num_experts=e if tight_capacity else None)
        assert ep.cap_rows == (t * min(k, -(-e // world)) if tight_capacity else t * k)
        ep.profile = True
        worst = 0.0
        for step in range(3):
            for l in range(L):
                x = acts(t - rank, h, torch.bfloat16, 700 + 10 * step + l + 100 * rank)  # ragged: ranks differ in T
                out = ep.forward(l, x, ws[l][0])
                ref = R.block_mixtral(x[None], ws[l][0], ws[l][1], top_k=k).out[0]
                worst = max(worst, float((out.float() - ref.float()).abs().max()))
                assert ep.last_form == ("variable" if (t - rank) * k > var_threshold else "fixed")
        ph = ep.phase_times_us()
        assert ph["calls"] == 3 * L and all(p in ph for p in ep.PHASES)
        q.put((rank, worst))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4])
def test_ep_gloo_matches_single_process_oracle(world):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, worst in res:
        assert worst == 0.0, f"rank {rank}: EP result differs from the oracle block by {worst}"


@pytest.mark.parametrize("world", [2, 4])
def test_ep_gloo_variable_split_exchange(world):
    """The prefill form: counts exchanged first, then exactly the routed rows with split sizes (forced here by a
    threshold of 0 pairs).  Ragged token counts per rank, same oracle equality."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q, 0)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, worst in res:
        assert worst == 0.0, f"rank {rank}: variable-split EP result differs from the oracle block by {worst}"


def test_ep_gloo_capacity_of_one_slot_per_token_with_one_expert_per_rank():
    """world size 8 with 8 experts: every rank owns ONE expert, so a token can send a rank at most one row — the fixed
    form carries tokens x 1 slots per peer (not tokens x K) and can never overflow."""
    world = 8
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q, 64, True, 12)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, worst in res:
        assert worst == 0.0, f"rank {rank}: EP result differs from the oracle block by {worst}"


# ---------------------------------------------------------------------------------------------------------------------
# Bootstrap of the engine-side transports (ExpertParallelMoE._try_peer_store): every step is local, its outcome is all-reduced,
# and ONE rank failing at ANY step must leave EVERY rank on the torch.distributed transport — never some ranks inside a
# transport the others did not enter.  A stand-in engine plays the C ABI (the real one is tests/test_gpu_ep_*.py).
# ---------------------------------------------------------------------------------------------------------------------
class _FakeEngine:
    PEER_BLOB_BYTES = 192

    def __init__(self, rank, world, fail_rank, fail_step):
        self.rank, self.world, self.fail_rank, self.fail_step = rank, world, fail_rank, fail_step
        self.attached_with, self.selected, self.uniform, self.forwards, self.released = None, None, None, 0, 0

    def _maybe_fail(self, step):
        if self.rank == self.fail_rank and step == self.fail_step:
            raise RuntimeError(f"injected failure at {step}")

    def ep_peer_export(self, cap_tokens):
        self._maybe_fail("export")
        return bytes([self.rank]) * self.PEER_BLOB_BYTES

    def ep_peer_attach(self, blobs):
        self._maybe_fail("attach")
        assert len(blobs) == self.world * self.PEER_BLOB_BYTES
        assert all(blobs[p * self.PEER_BLOB_BYTES] == p for p in range(self.world)), "blobs must arrive in rank order"
        self.attached_with = blobs

    def ep_peer_selftest(self):
        self._maybe_fail("selftest")
        return not (self.rank == self.fail_rank and self.fail_step == "selftest_false")

    def ep_peer_release(self):
        self.released += 1
        self.attached_with = None

    def ep_transport(self):
        return {"transport": "peer-store", "shared_device": False, "poll_in_kernels": True, "exchanges": 1}

    def ep_select_transport(self, name):
        self.selected = name

    def ep_set_uniform_tokens(self, on):
        self.uniform = on

    def ep_moe_forward(self, layer, x2, gate_w, out, batch_rows=1):
        self.forwards += 1


def _bootstrap_worker(rank, world, port, q, fail_rank, fail_step):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from ep_cpu_ops import OracleEpOps
        from moe_infinity_amd.ep import ExpertParallelMoE
        from oracle.synth import acts, make_weights

        h, f, e, k = 64, 64, 4, 2
        ws = make_weights("mixtral", h, f, e, 90, torch.bfloat16)
        ops = OracleEpOps([ws[1]], rank, world, k, e, h)
        ops.engine = _FakeEngine(rank, world, fail_rank, fail_step)
        # fail_step "one_rank_not_uniform": rank fail_rank does not promise equal token counts -> NO rank may take the broadcast form
        uniform = not (fail_step == "one_rank_not_uniform" and rank == fail_rank)
        ep = ExpertParallelMoE(ops, h, k, 4, torch.bfloat16, "cpu", num_experts=e, transport="peer-store", uniform_tokens=uniform)
        ep.forward(0, acts(2, h, torch.bfloat16, 91 + rank), ws[0])  # collective either way: native (fake) or torch transport (oracle ops)
        q.put((rank, ep.transport, ep.native, ep.native_note, ops.engine.selected, ops.engine.uniform, ops.engine.forwards, ops.engine.released))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,fail_rank,fail_step", [(2, -1, ""), (3, 1, "export"), (3, 2, "attach"), (2, 0, "selftest"), (3, 1, "selftest_false"),
                                                       (3, 2, "one_rank_not_uniform")])
def test_transport_bootstrap_is_all_or_nothing(world, fail_rank, fail_step):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_bootstrap_worker, args=(r, world, port, q, fail_rank, fail_step)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=180) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    want = "peer-store" if fail_rank < 0 or fail_step == "one_rank_not_uniform" else "torch"
    for rank, transport, native, note, selected, uniform, forwards, released in res:
        assert transport == want and native == (want != "torch"), (rank, transport, note)
        if want == "peer-store":
            # the exchange FORM is a group decision: one rank that does not promise uniform token counts switches the
            # broadcast form off on EVERY rank (advisor finding, round 4)
            assert selected == "peer-store" and uniform is (fail_rank < 0) and forwards == 1 and "self-test passed on every rank" in note
            assert released == 0
        else:  # every rank — the one that failed AND the ones that did not — stayed on torch.distributed and says why
            assert forwards == 0 and selected is None, (rank, note)
            assert released == 1, "a transport the group turned down gives its window and mappings back on every rank"
            assert ("failed" in note) or ("another rank" in note) or ("self-test" in note), note
