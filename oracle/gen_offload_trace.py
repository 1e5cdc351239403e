#!/usr/bin/env python3
"""Golden CALL TRACE of the reference's own caller of the boundary.  TEST INFRASTRUCTURE ONLY.

Runs, in this container (CPU), the reference's OWN code
    OffloadEngine._offload_state_dict          /root/reference/moe_infinity/runtime/model_offload.py:885-906
    OffloadEngine.setup_archer_hooks           :751-873   (incl. get_topology :638-749, gen_args_hook, register_expert)
    OffloadEngine._register_hooks_recursively  :908-991   (the begin/end pre/post-forward hooks of every module)
    SyncMixtralSparseMoeBlock.forward          /root/reference/moe_infinity/models/mixtral.py:40-118
    DistributedExpertExecutor.dispatch_local   /root/reference/moe_infinity/distributed/expert_executor.py:32-58
over a 2-layer toy decoder (Linear "attention" + the reference's Mixtral MoE block per layer, lm_head), against a
RECORDING stand-in of the pybind module `prefetch_op` (core/python/py_archer_prefetch.cpp:10-93): every call that
crosses the boundary is logged in order with its tensor ids / shapes, and performed functionally on the CPU so the
forward really runs.  Output: tests/golden/offload_engine_trace.json — the call sequence, the topology the reference's
get_topology produced, the name -> tensor-id map and the model output.

What the fixture is for: /root/reference cannot travel to the GPU box, so tests/test_gpu_dropin.py drives the REAL
moe-infinity_amd/prefetch_op.py on the GPU through a logging proxy and must reproduce THIS sequence call for call (and
the output); tests/test_ref_offload_trace_cpu.py re-runs this script's recording here and compares it with the
committed file, so the fixture cannot drift from the reference.

Stand-ins (everything else is reference code): the pybind module (the recorder below); HF MixtralBlockSparseTop2MLP (its
4.37 definition, removed in transformers 5 — same stand-in as oracle/gen_golden.py); moe_infinity package __init__s
that import CUDA-only / missing dependencies (stub packages, as gen_golden.import_reference); torch.cuda.device_count
-> 1 and `.to(0)` -> no-op (no GPU here).  The OffloadEngine object is built without its HF-bound __init__/
from_pretrained wrapper (model_offload.py:77-612): the attributes those set are filled in by hand below, citing lines.
"""
import importlib
import json
import os
import sys
import types

import torch
import torch.nn as nn

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
REF = "/root/reference"
OUT = os.path.join(ROOT, "tests", "golden", "offload_engine_trace.json")

L, H, F, E, K, V, T = 2, 64, 128, 8, 2, 96, 5
SEED = 4242


# ---- the toy model (names chosen so that get_topology's regexes see what they see in Mixtral: "layers.N." dense nodes,
# ---- "...experts" sparse nodes with the expert index third from the end, "lm_head")
def build_model(moe_block_cls, dtype=torch.bfloat16):
    cfg = types.SimpleNamespace(hidden_size=H, intermediate_size=F, num_local_experts=E, num_experts_per_tok=K, hidden_act="silu")

    class Layer(nn.Module):
        def __init__(self):
            super().__init__()
            self.attn = nn.Linear(H, H, bias=True)
            self.block_sparse_moe = moe_block_cls(cfg)

        def forward(self, x):
            x = x + self.attn(x)
            y, _ = self.block_sparse_moe(x)
            return x + y

    class Toy(nn.Module):
        def __init__(self):
            super().__init__()
            self.layers = nn.ModuleList([Layer() for _ in range(L)])
            self.lm_head = nn.Linear(H, V, bias=False)

        def forward(self, x):
            for layer in self.layers:
                x = layer(x)
            return self.lm_head(x)

    torch.manual_seed(SEED)
    m = Toy()
    with torch.no_grad():
        for n, p in m.named_parameters():
            p.copy_(torch.randn(p.shape) * (0.5 if "gate" in n else 0.08))
    return m.to(dtype)


def toy_input(dtype=torch.bfloat16):
    g = torch.Generator().manual_seed(SEED + 1)
    x = torch.randn(1, T, H, generator=g)
    return (x / x.pow(2).mean(-1, keepdim=True).sqrt()).to(dtype)


# ---- recording stand-in of the pybind module --------------------------------------------------------------------
class Recorder:
    def __init__(self):
        self.calls = []
        self.store = {}      # tensor id -> offloaded tensor
        self.ptr2id = {}     # data_ptr of the tensor a parameter currently holds -> tensor id
        self.topology = None

    def log(self, *c):
        self.calls.append(list(c))


class RecHandle:
    """prefetch_handle(prefix, device_memory_ratio) — py_archer_prefetch.cpp:12-80"""

    def __init__(self, rec):
        self.r = rec

    def is_tensor_offloaded(self, tid):
        return int(tid) in self.r.store

    def offload(self, tensor, tid):
        self.r.store[int(tid)] = tensor.detach().clone()
        self.r.log("offload", int(tid), list(tensor.shape), str(tensor.dtype))

    def register(self, tensor, tid):
        self.r.ptr2id[tensor.data_ptr()] = int(tid)
        self.r.log("register", int(tid))

    def set_topology(self, topo):
        self.r.topology = [[name, [[int(t) for t in ids] for ids in groups]] for name, groups in topo]
        self.r.log("set_topology", len(topo))

    def get_node_default_device(self, ids):
        self.r.log("get_node_default_device", [int(t) for t in ids])
        return 0

    def fetch_tensors(self, request_id, ids):
        self.r.log("fetch_tensors", int(request_id), [int(t) for t in ids])

    def begin(self, request_id, param):
        tid = self.r.ptr2id.pop(param.data.data_ptr())
        param.data = self.r.store[tid]
        self.r.ptr2id[param.data.data_ptr()] = tid
        self.r.log("begin", int(request_id), tid)

    def end(self, request_id, param):
        tid = self.r.ptr2id.pop(param.data.data_ptr())
        param.data = torch.zeros(1, dtype=param.dtype)
        self.r.ptr2id[param.data.data_ptr()] = tid
        self.r.log("end", int(request_id), tid)


class RecDispatcher:
    """expert_dispatcher(num_experts, num_layers, dtype, expert_type, num_threads) — py_archer_prefetch.cpp:84-92"""

    def __init__(self, rec):
        self.r, self.ids, self.q = rec, {}, []

    def register_expert(self, layer, expert, ids):
        self.ids[(int(layer), int(expert))] = [int(t) for t in ids]
        self.r.log("register_expert", int(layer), int(expert), [int(t) for t in ids])

    def set_inputs(self, hidden, mask):
        self.hidden, self.mask = hidden, mask
        self.r.log("set_inputs", list(hidden.shape), list(mask.shape), [int(v) for v in mask.reshape(-1, mask.shape[-1]).sum(0)])

    def set_expected_queue(self, n):
        self.r.log("set_expected_queue", int(n))

    def enqueue_expert(self, layer, expert, gpu, remote):
        self.q.append((int(layer), int(expert)))
        self.r.log("enqueue_expert", int(layer), int(expert), int(gpu), bool(remote))

    def wait_expert(self):
        res = []
        for layer, e in self.q:  # MixtralExpert: (silu(x w1^T) * (x w3^T)) w2^T, blob order w1 w2 w3 (expert_module.cpp:147-175)
            w1, w2, w3 = (self.r.store[t] for t in self.ids[(layer, e)])
            x = self.hidden[self.mask[:, e].bool()]
            y = torch.nn.functional.linear(torch.nn.functional.silu(torch.nn.functional.linear(x, w1)) * torch.nn.functional.linear(x, w3), w2)
            res.append((y, layer, e, 0))
        self.q = []
        self.r.log("wait_expert", len(res))
        return res


# ---- import the reference's files ---------------------------------------------------------------------------------
def import_reference_offload_engine():
    from oracle.gen_golden import import_reference

    mods = import_reference()  # stub packages + the blocks + distributed.expert_executor + memory.*
    mi = sys.modules["moe_infinity"]
    # names model_offload.py imports at module level (only referenced by the parts we do not run)
    common = types.ModuleType("moe_infinity.common")
    common.parse_expert_type = lambda config: 4
    sys.modules["moe_infinity.common"] = common
    dist_pkg = sys.modules["moe_infinity.distributed"]
    dist_pkg.DistributedExpertExecutor = mods["executor"].DistributedExpertExecutor
    mem = sys.modules["moe_infinity.memory"]
    mem.ExpertPrefetcher = importlib.import_module("moe_infinity.memory.expert_prefetcher").ExpertPrefetcher
    models = sys.modules["moe_infinity.models"]
    models.SyncMixtralSparseMoeBlock = mods["mixtral"].SyncMixtralSparseMoeBlock
    for n in ("DeepseekMoEBlock", "SyncArcticMoeBlock", "SyncGrokMoeBlock", "SyncNllbMoeSparseMLP", "SyncSwitchTransformersSparseMLP"):
        setattr(models, n, type(n, (), {}))
    models.apply_rotary_pos_emb = models.apply_rotary_pos_emb_deepseek = None  # runtime/hooks.py imports them
    ops = types.ModuleType("moe_infinity.ops"); ops.__path__ = []
    opb = types.ModuleType("moe_infinity.ops.op_builder"); opb.__path__ = []
    pf = types.ModuleType("moe_infinity.ops.op_builder.prefetch")
    pf.PrefetchBuilder = type("PrefetchBuilder", (), {})
    sys.modules.update({"moe_infinity.ops": ops, "moe_infinity.ops.op_builder": opb, "moe_infinity.ops.op_builder.prefetch": pf})
    utils = sys.modules["moe_infinity.utils"]
    hf = sys.modules["moe_infinity.utils.hf_config"]
    utils.parse_expert_dtype = getattr(hf, "parse_expert_dtype", None)
    utils.parse_expert_id = getattr(hf, "parse_expert_id", None)
    rt = types.ModuleType("moe_infinity.runtime"); rt.__path__ = [f"{REF}/moe_infinity/runtime"]
    sys.modules["moe_infinity.runtime"] = rt
    mi.runtime = rt
    import transformers.modeling_utils as tmu  # transformers 5 moved PretrainedConfig out of modeling_utils (4.37 exported it there)
    if not hasattr(tmu, "PretrainedConfig"):
        from transformers import PretrainedConfig

        tmu.PretrainedConfig = PretrainedConfig
    mo = importlib.import_module("moe_infinity.runtime.model_offload")
    return mo, mods


def record():
    """-> dict(calls, topology, name_id_map, output, out_shape)"""
    os.environ.setdefault("PYTHONDONTWRITEBYTECODE", "1")
    sys.dont_write_bytecode = True
    mo, mods = import_reference_offload_engine()
    Block = mods["mixtral"].SyncMixtralSparseMoeBlock
    rec = Recorder()
    handle, disp = RecHandle(rec), RecDispatcher(rec)
    model = build_model(Block)
    x = toy_input()

    eng = object.__new__(mo.OffloadEngine)                 # __init__ / init() are HF- and pybind-bound (model_offload.py:77-165)
    eng.name_id_map, eng.offload_set, eng.offload_exemption = {}, set(), set()   # :99-105, :78
    eng.forward_hooks, eng.backward_hooks = [], []                              # :102-103
    eng.param_id, eng.request_id = 0, 0                                          # class attributes :66-67
    eng.config = types.SimpleNamespace(model_type="mixtral", first_k_dense_replace=0)
    eng.model_name = "mixtral"                                                   # from_pretrained wrapper :330
    eng.archer_engine = handle                                                   # :143-145
    eng.expert_dispatcher = disp                                                 # :471-477
    eng.expert_executor = mo.DistributedExpertExecutor(archer_config=None)       # :159-161
    eng.expert_executor.set_expert_dispatcher(disp)                              # :546-548

    real_count, real_to = torch.cuda.device_count, torch.Tensor.to
    torch.cuda.device_count = lambda: 1                                          # dispatch_local: gpu_id = expert_id % device_count

    def to_nogpu(self, *a, **k):  # hooks move activations to device index 0 (:780-790): no GPU here
        if a and (isinstance(a[0], int) or (isinstance(a[0], (str, torch.device)) and "cuda" in str(a[0]))):
            return self
        return real_to(self, *a, **k)

    torch.Tensor.to = to_nogpu
    try:
        eng._offload_state_dict(model.state_dict(), {})                          # REFERENCE CODE
        for _, p in model.named_parameters():                                    # apply_to_model_decorator's effect (:183-193)
            p.data = torch.zeros(1, dtype=p.dtype)
        idx = 0
        for module in model.modules():                                           # :550-603
            if isinstance(module, Block):
                module.archer_engine = handle
                module.expert_executor = eng.expert_executor
                module.layer_id = idx
                idx += 1
        eng.setup_archer_hooks(model)                                            # REFERENCE CODE
        rec.log("forward", 0)
        with torch.no_grad():
            y = model(x)                                                         # the reference's hooks + block + dispatch_local run
            rec.log("forward", 1)
            y2 = model(x)
        assert torch.equal(y, y2)
    finally:
        torch.cuda.device_count, torch.Tensor.to = real_count, real_to
    return dict(calls=rec.calls, topology=rec.topology, name_id_map=eng.name_id_map,
                output=[float(v) for v in y.float().reshape(-1)], out_shape=list(y.shape),
                shapes=dict(L=L, H=H, F=F, E=E, K=K, V=V, T=T, seed=SEED))


def main():
    d = record()
    with open(OUT, "w") as f:
        json.dump(d, f)
    print(f"wrote {OUT}: {len(d['calls'])} boundary calls, topology of {len(d['topology'])} nodes")


if __name__ == "__main__":
    main()
