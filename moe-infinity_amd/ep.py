"""Expert parallelism across the GPUs of one node.

One process per GPU.  Expert e of every layer lives on rank (e % world_size) — the reference's
placement (core/model/model_topology.cpp:533-536, moe_infinity/distributed/expert_executor.py:49-54),
which the reference serves from ONE process with implicit P2P ``tensor.to(device)`` copies
(core/parallel/expert_dispatcher.cpp:284,405).  Here every rank routes its own tokens, and routed
rows travel with one all-to-all each way (RCCL over xGMI; torch.distributed backend "nccl"):

    route (local)  ->  pack rows by destination rank  ->  all_to_all  ->  grouped expert FFN on the
    owner  ->  all_to_all back  ->  deterministic combine (local)

Decode-sized exchanges use a fixed per-peer capacity (tokens*K rows: worst case every pair goes to one rank), so
they need no host-side size negotiation; prefill-sized exchanges move exactly the routed rows (variable split,
see ``ExpertParallelMoE``).  Every exchange row is H activations plus a
16-byte tail holding the row's expert id (-1 = padding), so payload and metadata cross the fabric
in ONE all-to-all per direction: at decode sizes the exchange is latency-bound (KBs per peer) and the
number of collectives is what matters.

``ops`` abstracts the four compute steps so the host logic here can be exercised on CPU with the
gloo backend (tests/test_ep_gloo.py supplies oracle-backed ops); the product ops are
``HipEpOps`` (HIP kernels through the C ABI).
"""
import os
from typing import Optional

import torch
import torch.distributed as dist


class HipEpOps:
    """The EP compute steps on the HIP engine (include/moeinf.h: moeinf_ep_*)."""

    def __init__(self, engine):
        self.engine = engine

    def route(self, layer, x2, gate_w):
        from .engine import FWD_ROUTE_ONLY

        self.engine.forward(layer, x2, gate_w, flags=FWD_ROUTE_ONLY)

    def route_pack(self, layer, x2, gate_w, send, counts, cap_rows):
        self.engine.ep_route_pack(layer, x2, gate_w, send, counts, cap_rows)

    def row_elems(self):
        return self.engine.ep_row_elems()

    def pack(self, x2, send, counts, cap_rows):
        self.engine.ep_pack(x2, send, counts, cap_rows)

    def pack_compact(self, x2, send, counts):
        self.engine.ep_pack_compact(x2, send, counts)

    def expert_ffn(self, layer, recv, y, cap_rows):
        self.engine.ep_expert_ffn(layer, recv, y, cap_rows)

    def expert_ffn_rows(self, layer, recv, y, nrows):
        self.engine.ep_expert_ffn_rows(layer, recv, y, nrows)

    def combine(self, x2, ret, out, cap_rows):
        self.engine.ep_combine(x2, ret, out, cap_rows)


class ExpertParallelMoE:
    """Two exchange forms, chosen per call from the number of (token, k) pairs:

    * fixed capacity (decode: pairs <= ``var_threshold``): every peer gets ``cap_rows`` row slots, ONE equal-split
      all-to-all per direction, no host round trip — the exchange is latency-bound (KBs per peer);
    * variable split (prefill): rows are packed compactly by destination, the per-destination counts are exchanged
      first (one tiny all-to-all + the only host read of the path), then exactly the routed rows travel
      (``all_to_all_single`` with split sizes) — at G ranks the fixed form would move G x the routed bytes.
    """

    PHASES = ("route_pack", "a2a_dispatch", "owner_ffn", "a2a_combine", "combine")

    TRANSPORTS = ("auto", "peer-store", "rccl", "torch")
    BOOT_TIMEOUT_MS = 3000  # per poll of the bootstrap self-test (MOEINF_EP_PEER_TIMEOUT_MS governs the exchanges after it)

    def __init__(self, ops, hidden: int, top_k: int, max_tokens: int, dtype: torch.dtype, device,
                 group: Optional[dist.ProcessGroup] = None, var_threshold: int = 64, num_experts: Optional[int] = None,
                 native: Optional[bool] = False, transport: Optional[str] = None, uniform_tokens: bool = False):
        """transport (decode-sized, fixed-capacity exchanges; prefill-sized ones always go through torch.distributed):
        "peer-store" = rows stored straight into the peers' windows, no collective (csrc/ep_peer.h); "rccl" = RCCL called from
        inside the engine; both = ONE host call per layer; "torch" = all_to_all_single, five host calls per layer; "auto" =
        the first of those three that passes its self-test on EVERY rank.  ``native`` is the older switch: False = "torch",
        None / True = "auto".  ``uniform_tokens``: the caller's promise that every rank passes the same token count to every
        forward (a decode loop does) — one-token forwards over the peer-store transport then take the broadcast form (four
        launches per layer; include/moeinf.h: moeinf_ep_set_uniform_tokens)."""
        self.ops = ops
        self.group = group
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self.hidden, self.top_k = hidden, top_k
        # row slots per peer of the fixed form: a token sends a rank at most one row per expert that rank owns, so
        # tokens * min(K, ceil(E / world)) can never overflow (Mixtral on 8 ranks: 1 slot per token, not K) — and needs
        # no agreement between ranks, unlike any capacity that CAN overflow (switching to the variable form would have to
        # be decided collectively: one more host-synchronised collective per layer)
        per_rank = top_k if num_experts is None else min(top_k, -(-num_experts // self.world))
        self.cap_rows = max_tokens * per_rank
        self.max_tokens = max_tokens
        self.var_threshold = var_threshold
        n = self.world * self.cap_rows
        mk = lambda *s, dt=dtype: torch.zeros(*s, dtype=dt, device=device)  # noqa: E731
        ld = ops.row_elems()  # H + 16-byte tail (expert id)
        self.send, self.recv = mk(n, ld), mk(n, ld)
        self.y, self.ret = mk(n, hidden), mk(n, hidden)
        self.send_counts, self.recv_counts = mk(self.world, dt=torch.int32), mk(self.world, dt=torch.int32)
        self.device = torch.device(device)
        # per-phase timers (bench.py --gpus N): events on the current stream around the five phases of a layer
        self._profile = False
        self._marks = []
        self.last_form = None
        # native transport: RCCL called from inside the engine, a whole layer = ONE host call (moeinf_ep_moe_forward)
        self.native = False
        self.transport = "torch"
        self.native_note = "not requested"
        if transport is None:
            transport = "torch" if native is False else "auto"
        if transport not in self.TRANSPORTS:
            raise ValueError(f"transport must be one of {self.TRANSPORTS}")
        notes = []
        if transport in ("auto", "peer-store"):
            if self._try_peer_store(max_tokens):
                self.native, self.transport = True, "peer-store"
            notes.append(f"peer-store: {self.native_note}")
        if not self.native and transport in ("auto", "rccl"):
            if self._try_native(max_tokens):
                self.native, self.transport = True, "rccl"
            notes.append(f"rccl: {self.native_note}")
        if notes:
            self.native_note = "; ".join(notes)
        if self.native:
            self.ops.engine.ep_select_transport(self.transport)
            # the promise selects the exchange FORM of one-token forwards, and the form must be the same on every rank:
            # it holds only if EVERY rank made it
            self.ops.engine.ep_set_uniform_tokens(self._agree(bool(uniform_tokens)))

    @property
    def profile(self):
        return self._profile

    @profile.setter
    def profile(self, on):
        self._profile = bool(on)
        if self.native:  # bit 1 = per-phase events of ep_moe_forward; bit 0 (per-kernel events) stays as the caller set it
            eng = self.ops.engine
            eng.set_profiling((getattr(eng, "_profiling", 0) & 1) | (2 if on else 0))

    # -- native transport -----------------------------------------------------------------------
    def _comm_device(self):
        """where the small bootstrap tensors live: RCCL moves device tensors, a host-side backend (gloo) host tensors"""
        return self.device if dist.get_backend(self.group) == "nccl" else torch.device("cpu")

    def _agree(self, ok: bool) -> bool:
        """True only if EVERY rank says ok (a path that some ranks take and others do not would dead-lock)"""
        t = torch.tensor([1 if ok else 0], dtype=torch.int32, device=self._comm_device())
        dist.all_reduce(t, op=dist.ReduceOp.MIN, group=self.group)
        return bool(t.item())

    def _try_peer_store(self, cap_tokens: int) -> bool:
        """Bootstrap the direct peer-store exchange (include/moeinf.h: moeinf_ep_peer_*): every rank allocates and exports its
        window (local), the blobs are all-gathered through the existing process group, every rank maps its peers (local), then a
        tagged self-test crosses every pair both ways.  The outcome of EVERY step is all-reduced, so either all ranks end up on
        this transport or none does.  Works with any process-group backend — the group only carries 192 bytes per rank — and
        between ranks that share one GPU."""
        eng = getattr(self.ops, "engine", None)
        if eng is None or not hasattr(eng, "ep_peer_export"):  # (an engine with this method IS the HIP engine: it only exists on a GPU)
            self.native_note = "ops are not the HIP engine"
            return False
        blob, ok = bytes(eng.PEER_BLOB_BYTES), True
        try:
            blob = eng.ep_peer_export(cap_tokens)
        except Exception as ex:  # noqa: BLE001
            ok = False
            self.native_note = f"window export failed: {ex}"
        if not self._agree(ok):
            if ok:
                self.native_note = "window export failed on another rank"
            self._release_peer_store(eng)
            return False
        dev = self._comm_device()
        mine = torch.tensor(list(blob), dtype=torch.uint8, device=dev)
        parts = [torch.empty_like(mine) for _ in range(self.world)]
        dist.all_gather(parts, mine, group=self.group)
        blobs = b"".join(bytes(p.cpu().tolist()) for p in parts)
        try:
            eng.ep_peer_attach(blobs)
        except Exception as ex:  # noqa: BLE001
            ok = False
            self.native_note = f"mapping the peers' windows failed: {ex}"
        if not self._agree(ok):
            if ok:
                self.native_note = "mapping the peers' windows failed on another rank"
            self._release_peer_store(eng)
            return False
        prev_timeout = None
        try:
            # bounded bootstrap: a peer that never publishes costs BOOT_TIMEOUT_MS per poll of the self-test, not the
            # exchange's own (long) timeout; the setting that was in force comes back below
            if hasattr(eng, "ep_peer_set_timeout_ms"):
                prev_timeout = eng.ep_peer_set_timeout_ms(self.BOOT_TIMEOUT_MS)
            ok = eng.ep_peer_selftest()
            if not ok:
                self.native_note = "self-test: wrong rows or a peer never published (timeout)"
        except Exception as ex:  # noqa: BLE001
            ok = False
            self.native_note = f"self-test failed: {ex}"
        finally:
            # whatever happens here, this rank must reach the collective _agree() below: the others are waiting in it
            try:
                if prev_timeout:
                    eng.ep_peer_set_timeout_ms(prev_timeout)
            except Exception as ex:  # noqa: BLE001
                ok = False
                self.native_note = f"restoring the exchange timeout failed: {ex}"
        if not self._agree(ok):
            if ok:
                self.native_note = "self-test failed on another rank"
            self._release_peer_store(eng)
            return False
        t = eng.ep_transport()
        self.native_note = ("rows stored straight into the peers' windows, no collective (self-test passed on every rank; "
                            + ("ranks share a GPU: one-wave wait kernels" if not t["poll_in_kernels"] else "consumer kernels poll their flags") + ")")
        return True

    def _release_peer_store(self, eng):
        """every rank took the same 'not this transport' branch (the verdicts are all-reduced): unmap the peers and free the
        window and its staging buffers — once nobody can still be storing into anybody's window"""
        try:
            dist.barrier(group=self.group)
            eng.ep_peer_release()
        except Exception as ex:  # noqa: BLE001
            self.native_note += f" (release: {ex})"

    def drop_native(self):
        """Collective: the group decided (outside: bench.py's probation) not to use the native transport this object
        bootstrapped — back to torch.distributed; a peer-store window is unmapped and freed on every rank."""
        if self.native and self.transport == "peer-store":
            self._release_peer_store(self.ops.engine)
        self.native, self.transport = False, "torch"

    def _try_native(self, cap_tokens: int) -> bool:
        """Bootstrap the engine's own RCCL communicator through the existing process group, then PROVE it: a tagged
        all-to-all must deliver the expected rows on every rank.  Any failure, on any rank, leaves every rank on the
        torch.distributed transport (the decision is collective)."""
        eng = getattr(self.ops, "engine", None)
        if eng is None or self.device.type != "cuda" or not hasattr(eng, "ep_comm_init"):
            self.native_note = "ops are not the HIP engine"
            return False
        if dist.get_backend(self.group) != "nccl":
            self.native_note = "process group is not RCCL (several ranks may share one GPU)"
            return False
        uid, ok = bytes(128), False
        try:
            ok = eng.ep_comm_available()       # every rank: librccl can be bound here
            if ok and self.rank == 0:
                uid = eng.ep_comm_unique_id()  # rank 0 only: the id carries its bootstrap address
        except Exception as ex:  # noqa: BLE001
            ok = False
            self.native_note = f"librccl not usable: {ex}"
        if not self._agree(ok):
            if ok:
                self.native_note = "librccl not usable on another rank"
            return False
        # the local half first (buffers, validation), agreed on BEFORE anyone enters the collective ncclCommInitRank
        try:
            eng.ep_comm_prepare(cap_tokens)
        except Exception as ex:  # noqa: BLE001
            ok = False
            self.native_note = f"exchange buffers: {ex}"
        if not self._agree(ok):
            if ok:
                self.native_note = "exchange buffers could not be made on another rank"
            return False
        t = torch.tensor(list(uid), dtype=torch.uint8, device=self.device)
        dist.broadcast(t, src=dist.get_global_rank(self.group, 0) if self.group is not None else 0, group=self.group)
        ok = True
        try:
            eng.ep_comm_init(bytes(t.cpu().tolist()), cap_tokens)
        except Exception as ex:  # noqa: BLE001
            ok = False
            self.native_note = f"ncclCommInitRank failed: {ex}"
        if not self._agree(ok):
            return False
        # self-test: segment p of rank r carries the value 16*r + p; after the exchange segment p must hold 16*p + r
        seg = 1024
        send = torch.empty(self.world * seg, dtype=torch.int32, device=self.device)
        for p in range(self.world):
            send[p * seg:(p + 1) * seg] = 16 * self.rank + p
        recv = torch.full_like(send, -1)
        try:
            eng.ep_all_to_all(send, recv)
            torch.cuda.synchronize(self.device)
            want = torch.cat([torch.full((seg,), 16 * p + self.rank, dtype=torch.int32) for p in range(self.world)])
            ok = bool(torch.equal(recv.cpu(), want))
            if not ok:
                self.native_note = "native all-to-all self-test delivered wrong rows"
        except Exception as ex:  # noqa: BLE001
            ok = False
            self.native_note = f"native all-to-all failed: {ex}"
        if not self._agree(ok):
            return False
        self.native_note = "RCCL inside the engine (self-test passed on every rank)"
        return True

    # -- timers ---------------------------------------------------------------------------------
    def _mark(self):
        if not self.profile or (self.native and len(self._marks) >= 6 * 4096):  # bounded when nobody collects them
            return
        if self.device.type == "cuda":
            ev = torch.cuda.Event(enable_timing=True)
            ev.record(torch.cuda.current_stream(self.device))
            self._marks.append(ev)
        else:
            import time

            self._marks.append(time.perf_counter())

    def phase_times_us(self):
        """Mean microseconds per layer call of each phase since profiling was switched on."""
        if self.native:
            p = self.ops.engine.ep_profile()
            c = max(1, p["calls"])
            self._marks = []  # (forwards that took the variable-split path left theirs here; they are not part of this report)
            return {"calls": p["calls"], **{ph: round(p[ph + "_ms"] * 1e3 / c, 2) for ph in self.PHASES},
                    "timed_by": "HIP events inside moeinf_ep_moe_forward"}
        if self.device.type == "cuda":
            torch.cuda.synchronize(self.device)
        n = len(self.PHASES) + 1
        calls = len(self._marks) // n
        tot = [0.0] * len(self.PHASES)
        for c in range(calls):
            m = self._marks[c * n:(c + 1) * n]
            for i in range(len(self.PHASES)):
                tot[i] += (m[i].elapsed_time(m[i + 1]) * 1e3) if self.device.type == "cuda" else (m[i + 1] - m[i]) * 1e6
        self._marks = []
        return {"calls": calls, **{p: round(t / max(1, calls), 2) for p, t in zip(self.PHASES, tot)}}

    # -- transport ------------------------------------------------------------------------------
    def _a2a(self, out, inp, out_splits=None, in_splits=None):
        """One all-to-all.  RCCL ("nccl") moves device buffers directly.  With a host-side backend (gloo) and device
        buffers — several ranks sharing ONE GPU, which RCCL refuses ("Duplicate GPU detected") — the rows are staged
        through host memory: same rows, same order, only the transport differs.  That is how the whole multi-process path
        (HIP kernels, exchange logic, two real processes) is tested on a one-GPU box."""
        if out.device.type == "cuda" and dist.get_backend(self.group) != "nccl":
            ho, hi = torch.empty(out.shape, dtype=out.dtype), inp.cpu()
            dist.all_to_all_single(ho, hi, output_split_sizes=out_splits, input_split_sizes=in_splits, group=self.group)
            out.copy_(ho)
            return
        dist.all_to_all_single(out, inp, output_split_sizes=out_splits, input_split_sizes=in_splits, group=self.group)

    # -- forward --------------------------------------------------------------------------------
    def forward(self, layer: int, x: torch.Tensor, gate_w: torch.Tensor, out: Optional[torch.Tensor] = None):
        shape = x.shape
        x2 = x.reshape(-1, shape[-1])
        if x2.shape[0] > self.max_tokens:
            raise ValueError("more tokens than the exchange buffers were sized for")
        if out is None:
            out = torch.empty_like(x2)
        if x2.shape[0] * self.top_k > self.var_threshold and self.world > 1:
            self._forward_variable(layer, x2, gate_w, out)
        elif self.native:
            self.last_form = "fixed"
            self.ops.engine.ep_moe_forward(layer, x2, gate_w, out)
        else:
            self._forward_fixed(layer, x2, gate_w, out)
        return out.reshape(shape)

    def _forward_fixed(self, layer, x2, gate_w, out):
        self.last_form = "fixed"
        self._mark()
        if hasattr(self.ops, "route_pack"):  # one host call; decode-sized: the router's own launch writes the send rows
            self.ops.route_pack(layer, x2, gate_w, self.send, None, self.cap_rows)
        else:
            self.ops.route(layer, x2, gate_w)
            self.ops.pack(x2, self.send, None, self.cap_rows)
        self._mark()
        # dispatch all-to-all: rows with their expert ids in the tail, equal splits of cap_rows per peer
        self._a2a(self.recv, self.send)
        self._mark()
        self.ops.expert_ffn(layer, self.recv, self.y, self.cap_rows)
        self._mark()
        # combine all-to-all: expert outputs return to the rows' home rank, same row positions
        self._a2a(self.ret, self.y)
        self._mark()
        self.ops.combine(x2, self.ret, out, self.cap_rows)
        self._mark()

    def _forward_variable(self, layer, x2, gate_w, out):
        self.last_form = "variable"
        self._mark()
        self.ops.route(layer, x2, gate_w)
        self.ops.pack_compact(x2, self.send, self.send_counts)
        # counts first (world int32 each way), then the one host read of this form: the split sizes
        self._a2a(self.recv_counts, self.send_counts)
        sc = self.send_counts.cpu().tolist()
        rc = self.recv_counts.cpu().tolist()
        n_out, n_in = sum(sc), sum(rc)
        if n_in > self.recv.shape[0]:
            raise ValueError("received more rows than the exchange buffers hold")
        self._mark()
        self._a2a(self.recv[:n_in], self.send[:n_out], rc, sc)
        self._mark()
        self.ops.expert_ffn_rows(layer, self.recv, self.y, n_in)
        self._mark()
        self._a2a(self.ret[:n_out], self.y[:n_in], sc, rc)
        self._mark()
        self.ops.combine(x2, self.ret, out, 0)
        self._mark()
