"""``prefetch_op`` — drop-in for the reference's pybind module ``moe_infinity.ops.prefetch.prefetch_op``
(core/python/py_archer_prefetch.cpp:10-93): the two classes ``prefetch_handle`` and ``expert_dispatcher`` with the
reference's constructor signatures, method names, argument meaning and results, on top of the C ABI
(include/moeinf.h).  The reference's Python callers run against it UNMODIFIED:

    moe_infinity/runtime/model_offload.py      OffloadEngine: ``prefetch_lib.prefetch_handle(path, ratio)``,
                                               ``prefetch_lib.expert_dispatcher(E, L, dtype, expert_type, threads)``,
                                               offload / register / set_topology / begin / end / fetch_tensors ...
    moe_infinity/distributed/expert_executor.py  dispatch_local: set_inputs / set_expected_queue / enqueue_expert / wait_expert
    moe_infinity/memory/expert_prefetcher.py     replace_cache_candidates / enqueue_prefetch / get_node_default_device

Who owns what (as in the reference, SURVEY.md section 8b "Ownership"):
  * tensors handed to ``register`` are BORROWED: the handle keeps, per tensor id, a view of where the data lives now
    (a slice of the node's device slab, or a 1-element placeholder) and ``begin`` re-points the caller's parameter at
    it (``ArcherTensorHandle::SetTensor``), ``end`` parks the parameter on a 1-element CPU tensor
    (archer_prefetch_handle.cpp:83-180); tensor identity is ``data_ptr`` -> id, hence ``update_tensor_map``;
  * EXPERT nodes (stages with more than one node, model_topology.cpp:416-417) live in the HIP engine: pinned host arena
    (or the offload directory when the arena is capped) + HBM slots; ``expert_dispatcher`` runs them with ONE grouped
    launch per FFN stage (moeinf_dispatch_mask);
  * DENSE nodes are placed on the device by ``set_topology`` ("Moving dense parameters to GPU",
    model_topology.cpp:518-530) as one device slab per node, filled disk -> pinned pieces -> HBM by
    moeinf_store_get_device; over the reference's dense cache limit (0.7 x device memory, cuda_utils.h:33) the
    earliest layers not in use are dropped first and re-read on their next ``begin`` (RemoveCachedDenseNode,
    task_scheduler.cpp:319-378).
One handle per process (the reference creates six process-wide singletons in the handle's constructor,
archer_prefetch_handle.cpp:18-27); ``expert_dispatcher`` attaches to it.

Several GPUs from ONE process (the reference's own multi-GPU form: sparse nodes are dealt round-robin over the visible
devices, model_topology.cpp:533-536; ``dispatch_local`` passes ``gpu_id = expert_id % total_gpus``,
moe_infinity/distributed/expert_executor.py:49-54, and ``Enqueue`` queues the expert on that GPU unless it is resident
elsewhere, core/parallel/expert_dispatcher.cpp:135-137; inputs travel with ``tensor.to(device)`` and outputs come back
to the hidden states' device, :284,405): the handle keeps ONE HIP engine per device of ``configure(devices=[...])``
(default: every visible device when the process is not a torch.distributed rank, else the current device only).
``enqueue_expert(layer, expert, gpu_id, remote)`` runs the expert on the engine of ``gpu_id`` (or where it is resident);
``wait_expert`` issues one grouped launch per FFN stage PER DEVICE, all devices concurrently, and returns the rows on the
hidden states' device.  ``devices=[0, 0]`` puts two engines on one GPU — how the multi-device path is tested on a
one-GPU box.  Dense nodes stay on the handle's own device (``device_id``).  For one PROCESS per GPU with routed rows
exchanged between ranks (RCCL / peer stores) use ``ExpertParallelMoE`` (ep.py).

Additions that have no counterpart in the reference's signatures are set through ``configure()`` before the handle
is created: ``device_memory_bytes`` (explicit expert-cache budget), ``host_memory_bytes``, ``cache_policy``,
``max_tokens`` (initial workspace; grows on demand) and ``top_k`` (workspace rows per token for dense masks).
"""
import os
from concurrent.futures import ThreadPoolExecutor
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch

from . import config as Cf
from .engine import MoEEngine
from .offload_store import TORCH_DTYPE, OffloadStore

_ALIGN = 4096  # kAioAlignment
_CURRENT: Optional["prefetch_handle"] = None
_OPTIONS = dict(device_memory_bytes=0, host_memory_bytes=0, cache_policy="lfu_incache", max_tokens=256, top_k=0, device_id=None,
                dense_cache_fraction=0.7, devices=None)


def configure(**kw):
    """engine options the reference's constructors have no argument for (see module docstring)"""
    for k, v in kw.items():
        if k not in _OPTIONS:
            raise KeyError(k)
        _OPTIONS[k] = v


class _Node:
    __slots__ = ("ids", "sparse", "stage", "index", "byte_size", "offsets", "slab", "in_use", "expert", "visit", "hit", "prefetch")

    def __init__(self, ids, sparse, stage, index):
        self.ids, self.sparse, self.stage, self.index = list(ids), sparse, stage, index
        self.byte_size, self.offsets = 0, {}
        self.slab = None       # torch.uint8 device tensor while the node is on the device (dense nodes)
        self.in_use = False    # between begin and end (exec_queue_ of the reference)
        self.expert = None     # (layer, expert) once expert_dispatcher.register_expert named it
        self.visit = self.hit = self.prefetch = 0


class prefetch_handle:
    def __init__(self, prefix: str, device_memory_ratio: float):
        global _CURRENT
        if not torch.cuda.is_available():
            raise RuntimeError("prefetch_op needs a visible MI355X (no CPU fallback)")
        self.prefix = prefix
        self.device_memory_ratio = float(device_memory_ratio)
        self.device_id = torch.cuda.current_device() if _OPTIONS["device_id"] is None else int(_OPTIONS["device_id"])
        self.device = torch.device("cuda", self.device_id)
        self.store = OffloadStore(prefix)
        self._index_initialized = len(self.store) > 0  # ArcherTensorHandle ctor: an index file existed
        self._dirty = False
        self._view: Dict[int, torch.Tensor] = {}     # tensor id -> where the data lives now
        self._ptr_to_id: Dict[int, int] = {}
        self._nodes: List[_Node] = []
        self._node_of: Dict[int, _Node] = {}
        self._acquired: Dict[int, set] = {}           # id(node) -> tensor ids still held (node_id_to_tensor_ids_)
        self._last_node: Optional[_Node] = None
        self._last_layer = 0
        self._child_visit = None                      # set_trace / get_trace
        # one engine per expert device ("slot" = index into self.devices = the reference's gpu_id)
        if _OPTIONS["devices"] is not None:
            self.devices = [int(d) for d in _OPTIONS["devices"]]
        elif _OPTIONS["device_id"] is None and int(os.environ.get("WORLD_SIZE", "1")) <= 1:
            self.devices = list(range(torch.cuda.device_count()))
        else:
            self.devices = [self.device_id]
        if not self.devices or any(d < 0 or d >= torch.cuda.device_count() for d in self.devices):
            raise RuntimeError(f"configure(devices={self.devices}): not visible devices (device_count {torch.cuda.device_count()})")
        self.engines: List[Optional[MoEEngine]] = [None] * len(self.devices)
        self._slot_of: Dict[Tuple[int, int], int] = {}    # (layer, expert) -> slot whose engine holds the expert
        self._expert_ids: Dict[Tuple[int, int], List[int]] = {}
        self._dispatcher = None
        self._dense_bytes = 0
        self._cleaned = False
        _CURRENT = self

    # ---- index / registration -----------------------------------------------------------------------------
    def offload(self, tensor: torch.Tensor, tensor_id: int):
        """OffloadTensor (archer_prefetch_handle.cpp:229-237): payload -> archer_param_<n>, entry -> archer_index"""
        self.store.offload(tensor, int(tensor_id))
        self._dirty = True

    def _flush(self):
        if self._dirty:
            self.store.flush()
            self._dirty = False

    def is_tensor_offloaded(self, tensor_id: int) -> bool:
        return self.store.is_tensor_offloaded(int(tensor_id))

    def is_tensor_index_initialized(self) -> bool:
        return self._index_initialized

    def register(self, tensor: torch.Tensor, tensor_id: Optional[int] = None):
        """RegisterTensor(tensor, id) (archer_tensor_handle.cpp:140-150); the one-argument overload only logs"""
        if tensor_id is None:
            return
        tensor_id = int(tensor_id)
        if not self.store.is_tensor_offloaded(tensor_id):
            raise RuntimeError(f"Tensor not found {tensor_id}")  # DLOG_FATAL in the reference
        self._ptr_to_id[tensor.data_ptr()] = tensor_id
        self._view[tensor_id] = tensor

    def update_tensor_map(self, old_data_ptr: int, new_data_ptr: int):
        if old_data_ptr not in self._ptr_to_id:
            raise RuntimeError(f"Tensor {old_data_ptr:#x} not found in tensor_to_id_")
        self._ptr_to_id[new_data_ptr] = self._ptr_to_id.pop(old_data_ptr)

    def _tensor_id(self, tensor: torch.Tensor) -> int:
        try:
            return self._ptr_to_id[tensor.data_ptr()]
        except KeyError:
            raise RuntimeError(f"Tensor not found {tensor.data_ptr():#x}") from None

    # ---- topology ------------------------------------------------------------------------------------------
    def set_topology(self, topology: Sequence[Tuple[str, Sequence[Sequence[int]]]]):
        """InitializeTopology (model_topology.cpp:402-548): one stage per entry, one node per id group; a stage with
        more than one node is sparse (experts).  Dense nodes go to the device now, expert nodes stay on the host /
        disk tier until dispatched."""
        self._flush()
        self._nodes, self._node_of = [], {}
        for stage, (_name, groups) in enumerate(topology):
            sparse = len(groups) > 1
            for index, ids in enumerate(groups):
                n = _Node(ids, sparse, stage, index)
                off = 0
                for tid in n.ids:
                    m = self.store.meta(int(tid))
                    if m is None:
                        raise RuntimeError(f"Tensor {tid} not found in tensor index")
                    n.offsets[int(tid)] = (off, m["nbytes"], tuple(m["shape"]), TORCH_DTYPE[m["scalar_type"]])
                    off += (m["nbytes"] + _ALIGN - 1) // _ALIGN * _ALIGN
                    self._node_of[int(tid)] = n
                n.byte_size = off
                self._nodes.append(n)
        for n in self._nodes:
            if n.sparse:
                for tid in n.ids:  # "Moving sparse parameters to CPU": the engine's tiers hold them
                    self._park(int(tid))
            else:
                self._to_device(n)

    def _park(self, tid: int):
        """the index's tensor for this id no longer aliases engine memory (SetModuleDisk, model_topology.cpp:630-642).
        The data_ptr -> id map is NOT touched: it tracks the CALLER'S parameter, which only begin / end /
        update_tensor_map re-point (archer_tensor_handle.cpp:158-187)."""
        if tid in self._view:
            t = self._view[tid]
            t.data = torch.zeros(1, dtype=t.dtype)

    def _dense_limit(self) -> int:
        _free, total = torch.cuda.mem_get_info(self.device)
        return int(total * _OPTIONS["dense_cache_fraction"])  # DEVICE_CACHE_LIMIT, cuda_utils.h:33

    def _to_device(self, n: _Node):
        if n.slab is not None:
            return
        # RemoveCachedDenseNode: over the limit, drop the earliest layers that are not in use
        limit = self._dense_limit()
        if self._dense_bytes + n.byte_size > limit:
            for v in sorted((v for v in self._nodes if v.slab is not None and not v.sparse and not v.in_use and v is not n),
                            key=lambda v: v.stage):
                self._evict(v)
                if self._dense_bytes + n.byte_size <= limit:
                    break
        n.slab = torch.empty(max(n.byte_size, 1), dtype=torch.uint8, device=self.device)
        for tid in n.ids:
            off, nbytes, shape, dt = n.offsets[int(tid)]
            if nbytes:
                self.store.load_to_device(int(tid), n.slab[off:off + nbytes])
            view = n.slab[off:off + nbytes].view(dt).reshape(shape)
            reg = self._view.get(int(tid))
            if reg is not None:
                reg.data = view  # SetModuleCudaMemoryFromCPU (model_topology.cpp:677-700): the index's tensor aliases the slab
            else:
                self._view[int(tid)] = view
        self._dense_bytes += n.byte_size

    def _evict(self, n: _Node):
        if n.slab is None:
            return
        for tid in n.ids:
            self._park(int(tid))
        n.slab = None
        self._dense_bytes -= n.byte_size

    def _node(self, tensor_id: int) -> _Node:
        try:
            return self._node_of[int(tensor_id)]
        except KeyError:
            raise RuntimeError(f"Tensor {tensor_id} not found in tensor id to node map") from None

    # ---- acquire / release (dense modules' forward hooks) --------------------------------------------------
    # ---- the engines ---------------------------------------------------------------------------------------
    @property
    def engine(self) -> Optional[MoEEngine]:
        """the first engine (single-device callers and tests)"""
        return self.engines[0]

    def _live_engines(self) -> List[MoEEngine]:
        return [e for e in self.engines if e is not None]

    def _engine_of(self, expert: Tuple[int, int]) -> Optional[MoEEngine]:
        s = self._slot_of.get(expert)
        return None if s is None else self.engines[s]

    def _resident(self, expert: Tuple[int, int]) -> bool:
        eng = self._engine_of(expert)
        return eng is not None and eng.is_resident(*expert)

    def begin(self, request_id: int, tensor: torch.Tensor):
        """AcquireTensor (archer_prefetch_handle.cpp:83-130)"""
        tid = self._tensor_id(tensor)
        old = tensor.data_ptr()
        n = self._node(tid)
        if not self._acquired.get(id(n)):
            self._acquired[id(n)] = set(int(t) for t in n.ids)
            n.visit += 1
            if n.slab is not None or (n.sparse and n.expert and self._resident(n.expert)):
                n.hit += 1
            n.in_use = True
            if not (n.sparse and n.expert):
                self._to_device(n)  # StartExec + wait: on return the node is on its device
        view = self._view[tid]
        tensor.data = view if view.dtype == tensor.dtype else view.to(tensor.dtype)  # ArcherTensorHandle::SetTensor
        self._ptr_to_id.pop(old, None)
        self._ptr_to_id[tensor.data_ptr()] = tid

    def end(self, request_id: int, tensor: torch.Tensor):
        """ReleaseTensor (archer_prefetch_handle.cpp:131-180)"""
        tid = self._tensor_id(tensor)
        old = tensor.data_ptr()
        n = self._node(tid)
        if id(n) not in self._acquired:
            return  # "Node not found in node_id_to_tensor_ids_" (logged, ignored)
        layer = n.stage
        if self._last_node is not None and layer != self._last_layer and self._acquired.get(id(self._last_node)):
            self._acquired[id(self._last_node)].clear()
            self._last_node.in_use = False  # StopExec(last_node)
        self._last_layer, self._last_node = layer, n
        self._acquired[id(n)].discard(tid)
        if not self._acquired[id(n)]:
            n.in_use = False  # StopExec(node)
            del self._acquired[id(n)]
        tensor.data = torch.zeros(1, dtype=tensor.dtype)
        self._ptr_to_id.pop(old, None)
        self._ptr_to_id[tensor.data_ptr()] = tid

    def fetch_tensors(self, request_id: int, tensor_ids: Sequence[int]):
        """FetchTensors -> FetchExec (archer_prefetch_handle.cpp:220-227, task_scheduler.cpp:44-80): bring the node of
        every id to its device at the most urgent level"""
        for tid in tensor_ids:
            n = self._node(tid)
            if n.sparse and n.expert and self._engine_of(n.expert) is not None:
                self._engine_of(n.expert).prefetch(n.expert[0], [n.expert[1]], scores=[1.0])
            elif not n.sparse:
                self._to_device(n)

    def prefetch_tensors(self, request_id: int, tensor_ids: Sequence[int]):
        """A no-op in the reference too (archer_prefetch_handle.cpp:182-193)."""
        return None

    def set_tensor_device(self, tensor: torch.Tensor, device):
        """SetTensorDevice (archer_prefetch_handle.cpp:325-345): a fresh device copy owned by the tensor"""
        tensor.data = tensor.data.to(device).clone()

    # ---- queries -------------------------------------------------------------------------------------------
    def get_node_default_device(self, tensor_ids: Sequence[int]) -> int:
        """sparse nodes: dealt round-robin over the expert devices in node order (model_topology.cpp:533-536)"""
        n = self._node(tensor_ids[0])
        if n.sparse:
            return self.devices[(self._slot_of.get(n.expert) if n.expert in self._slot_of else n.index % len(self.devices))]
        return self.device_id

    def get_node_device(self, tensor_ids: Sequence[int]) -> int:
        n = self._node(tensor_ids[0])
        if n.sparse and n.expert and self._engine_of(n.expert) is not None:
            return self.devices[self._slot_of[n.expert]] if self._resident(n.expert) else -1
        return self.device_id if n.slab is not None else -1

    def is_tensor_on_device(self, tensor_or_id) -> bool:
        tid = self._tensor_id(tensor_or_id) if isinstance(tensor_or_id, torch.Tensor) else int(tensor_or_id)
        return self.get_node_device([tid]) >= 0

    def get_hit_rate(self) -> torch.Tensor:
        """GetHitRate (archer_prefetch_handle.cpp:281-297; columns: model_topology.cpp:253-263), one row per node in
        topology order.  Experts take their counters from the engine."""
        # columns (model_topology.cpp:253-263): visit_cnt, gpu_visit_cnt, cpu_visit_cnt, hit_cnt, gpu_hit_cnt, cpu_hit_cnt,
        # len(tensor_ids), prefetch_cnt, unused_count, io_state, is_sparse.  What the reference's code can actually put
        # there: every execution targets the GPU, so gpu_visit_cnt == visit_cnt (task_scheduler.cpp:140,154-156) and
        # cpu_visit_cnt stays 0 (its only increment is commented out, :147-151); cpu_hit_cnt needs a CPU destination
        # (:179-181: never requested); io_state is reset to NODE_STATE_NONE by the getter itself (:250).  unused_count is
        # real: evictions of a speculatively fetched node nobody visited (:304) — an engine counter for experts.
        rows = np.zeros((len(self._nodes), 11), np.int64)
        cs = [None if e is None else e.expert_counters() for e in self.engines]
        for i, n in enumerate(self._nodes):
            v, h, p, unused = n.visit, n.hit, n.prefetch, 0
            c = cs[self._slot_of[n.expert]] if (n.sparse and n.expert in self._slot_of) else None
            if c is not None:
                v, h, _m, p = (int(x) for x in c[n.expert[0], n.expert[1], :4])
                unused = int(c[n.expert[0], n.expert[1], 6])
            rows[i] = [v, v, 0, h, h, 0, len(n.ids), p, unused, 0, int(n.sparse)]
        return torch.from_numpy(rows)

    def set_trace(self, trace: torch.Tensor):
        """SetTrace (archer_prefetch_handle.cpp:299-308 -> SetChildVisitCounts, model_topology.cpp:336-377): child-visit
        counts [L-1, E, E]; a wrong shape is logged and ignored there — here it raises"""
        if trace.dim() != 3 or not trace.is_contiguous() or trace.device.type != "cpu":
            raise ValueError("Trace should be a contiguous 3D tensor on CPU")
        layers = sum(1 for s in {n.stage for n in self._nodes if n.sparse})
        experts = max((n.index + 1 for n in self._nodes if n.sparse), default=0)
        if self._nodes and tuple(trace.shape) != (max(layers - 1, 0), experts, experts):
            raise ValueError(f"visit_counts size {tuple(trace.shape)} not equal to ({layers - 1}, {experts}, {experts})")
        self._child_visit = trace.to(torch.int64).clone()

    def get_trace(self) -> torch.Tensor:
        """GetTrace (archer_prefetch_handle.cpp:263-279): the child-visit counts last set (zeros before)"""
        if self._child_visit is not None:
            return self._child_visit.clone()
        layers = len({n.stage for n in self._nodes if n.sparse})
        experts = max((n.index + 1 for n in self._nodes if n.sparse), default=0)
        return torch.zeros((max(layers - 1, 0), experts, experts), dtype=torch.int64)

    # ---- expert cache control (memory/expert_prefetcher.py) ------------------------------------------------
    def _experts(self, tensor_ids: Sequence[int]) -> List[Tuple[int, int]]:
        seen, out = set(), []
        for tid in tensor_ids:
            n = self._node(tid)
            if n.expert is None:
                raise RuntimeError(f"tensor id {tid} does not belong to a registered expert")
            if n.expert not in seen:
                seen.add(n.expert)
                out.append(n.expert)
        return out

    def replace_cache_candidates(self, tensor_ids: Sequence[int]):
        """every engine gets the candidates it holds (an engine without any drops its protected set)"""
        per = {s: [] for s in range(len(self.engines))}
        for ex in self._experts(tensor_ids):
            if ex in self._slot_of:
                per[self._slot_of[ex]].append(ex)
        for s, eng in enumerate(self.engines):
            if eng is not None:
                eng.protect(per[s])

    def enqueue_prefetch(self, tensor_id: int, gpu_id: int = 0):
        """EnqueuePrefetch(tensor_id, gpu_id) (archer_prefetch_handle.cpp:195-206): the reference's caller passes the node's
        default device (memory/expert_prefetcher.py:56-59) — the device whose engine holds the expert here."""
        layer, expert = self._experts([tensor_id])[0]
        eng = self._engine_of((layer, expert))
        if eng is not None:
            eng.prefetch(layer, [expert])

    def clean_up_resources(self):
        global _CURRENT
        if self._cleaned:
            return
        self._flush()
        for n in self._nodes:
            n.slab = None
        for s, eng in enumerate(self.engines):
            if eng is not None:
                eng.close()
                self.engines[s] = None
        self.store.close()
        self._cleaned = True
        if _CURRENT is self:
            _CURRENT = None

    def __del__(self):
        try:
            self.clean_up_resources()
        except Exception:
            pass


_ROUTER_OF = {Cf.EXPERT_SWITCH: Cf.ROUTER_SWITCH, Cf.EXPERT_SWITCH_GATED: Cf.ROUTER_SWITCH, Cf.EXPERT_NLLB: Cf.ROUTER_NLLB, Cf.EXPERT_FSGPT: Cf.ROUTER_NLLB,
              Cf.EXPERT_MIXTRAL: Cf.ROUTER_MIXTRAL, Cf.EXPERT_DEEPSEEK: Cf.ROUTER_DEEPSEEK}


def slot_for_gpu(devices: Sequence[int], gpu_id: int, home: Optional[int]) -> int:
    """Which of the handle's engines serves an ``enqueue_expert(..., gpu_id)``.

    The reference's dispatch_local names ``gpu_id = expert_id % torch.cuda.device_count()`` (expert_executor.py:49-54), a
    CUDA ordinal of ONE process that drives every GPU.  This handle drives ``devices`` (configure(devices=[...]); a single
    device under WORLD_SIZE > 1 or configure(device_id=...)):
      * distinct devices: ``gpu_id`` is looked up as an ordinal (devices=[2, 3], gpu_id 3 -> engine 1);
      * the same device listed several times (tests: several engines on the one GPU of the box): ``gpu_id`` is the
        engine's index;
      * anything else — a GPU the process can see but this handle does not drive — is the expert's HOME engine (where
        register_expert dealt it, expert % len(devices), model_topology.cpp:533-536), never an error: the unmodified
        dispatch_local must keep working when device_count() > len(devices)."""
    devices = list(devices)
    fallback = home if home is not None else (gpu_id % len(devices) if gpu_id >= 0 else 0)
    if len(set(devices)) == len(devices):
        return devices.index(gpu_id) if gpu_id in devices else fallback
    return gpu_id if 0 <= gpu_id < len(devices) else fallback


class expert_dispatcher:
    """expert_dispatcher(num_experts, num_layers, dtype, expert_type, num_threads)
    (py_archer_prefetch.cpp:84-92, core/parallel/expert_dispatcher.h:27-137).  ``num_threads`` sized the reference's
    exec-thread pool; here every wait_expert() is ONE grouped launch per FFN stage, so it is accepted and unused."""

    def __init__(self, num_experts: int, num_layers: int, dtype: int, expert_type: int, num_threads: int = 1):
        if _CURRENT is None:
            raise RuntimeError("create prefetch_handle(prefix, device_memory_ratio) first (the reference's dispatcher "
                               "reads the handle's process-wide singletons too)")
        self.handle = _CURRENT
        self.num_experts, self.num_layers, self.dtype, self.expert_type = int(num_experts), int(num_layers), int(dtype), int(expert_type)
        self.handle._dispatcher = self
        self._queue, self._expected, self._hidden, self._mask = [], 0, None, None
        self._placed = set()  # (layer, expert, slot): the blob is in that engine's host arena

    def _engine(self, slot: int, tensor_ids: Sequence[int]) -> MoEEngine:
        """the engine of expert device ``slot`` (created on first use: the dispatcher's constructor carries no shapes)"""
        h = self.handle
        if h.engines[slot] is None:
            m = h.store.meta(int(tensor_ids[0]))  # first tensor of every expert type is [F, H]
            if m is None or len(m["shape"]) != 2:
                raise RuntimeError(f"tensor {tensor_ids[0]} is not a [F, H] matrix in the offload index")
            f, hid = int(m["shape"][0]), int(m["shape"][1])
            rk = _ROUTER_OF.get(self.expert_type, Cf.ROUTER_MIXTRAL)
            # The dispatcher's constructor carries no top-k (py_archer_prefetch.cpp:84-85) and this path never runs the
            # engine's router (the caller's Python router hands over a dense mask), so K only SIZES the workspace
            # (max_tokens * K rows).  Without configure(top_k=...) the engine's own limit min(8, E) is used: an upper
            # bound for every supported model (K <= 8), i.e. more workspace, never too little.
            k = {Cf.ROUTER_SWITCH: 1, Cf.ROUTER_NLLB: 2}.get(rk, _OPTIONS["top_k"] or min(8, self.num_experts))
            # engines that share a physical device (devices=[0, 0], tests) share its budget
            share = h.devices.count(h.devices[slot])
            cfg = Cf.EngineConfig(num_layers=self.num_layers, num_experts=self.num_experts, expert_type=self.expert_type,
                                  hidden=hid, inter=f, top_k=k, router_kind=rk, dtype=self.dtype, device_id=h.devices[slot],
                                  device_memory_ratio=h.device_memory_ratio / share, device_memory_bytes=int(_OPTIONS["device_memory_bytes"]) // share,
                                  host_memory_bytes=int(_OPTIONS["host_memory_bytes"]) // len(h.devices),
                                  policy=Cf.POLICY_LRU if _OPTIONS["cache_policy"] == "lru" else Cf.POLICY_LFU_INCACHE,
                                  max_tokens=int(_OPTIONS["max_tokens"]))
            with torch.cuda.device(h.devices[slot]):  # (the C side selects its device per call; torch's current device comes back)
                h.engines[slot] = MoEEngine(cfg)
        return h.engines[slot]

    def _place(self, layer: int, expert: int, slot: int) -> MoEEngine:
        """the expert's blob goes to the pinned arena of ``slot``'s engine (disk -> host), once per (expert, slot)"""
        h = self.handle
        ids = h._expert_ids[(layer, expert)]
        eng = self._engine(slot, ids)
        if (layer, expert, slot) not in self._placed:
            h._flush()
            with torch.cuda.device(h.devices[slot]):
                h.store.register_expert(eng, layer, expert, ids)
            self._placed.add((layer, expert, slot))
        return eng

    def register_expert(self, layer_idx: int, expert_idx: int, tensor_ids: Sequence[int]):
        """RegisterExpert (expert_dispatcher.cpp:160-173): every id must belong to ONE node of the topology.  The expert is
        placed on its default device: dealt round-robin over the expert devices (model_topology.cpp:533-536) — the same
        device dispatch_local will name (expert_id % total_gpus, expert_executor.py:51)."""
        h = self.handle
        ids = [int(t) for t in tensor_ids]
        nodes = {id(h._node(t)) for t in ids}
        if len(nodes) != 1:
            raise RuntimeError(f"RegisterExpert: tensor_id has multiple nodes {ids}")
        key = (int(layer_idx), int(expert_idx))
        h._expert_ids[key] = ids
        slot = key[1] % len(h.devices)
        h._slot_of[key] = slot
        self._place(key[0], key[1], slot)
        h._node(ids[0]).expert = key

    def set_inputs(self, hidden_states: torch.Tensor, router_mask: torch.Tensor):
        self._hidden = hidden_states.reshape(-1, hidden_states.shape[-1]).contiguous()
        self._mask = router_mask.reshape(-1, router_mask.shape[-1])

    def set_expected_queue(self, expected_pending: int):
        self._expected = int(expected_pending)

    def enqueue_expert(self, layer_idx: int, expert_idx: int, gpu_id: int = 0, remote: bool = False):
        """EnqueueExpert -> Enqueue (expert_dispatcher.cpp:108-158): the expert runs on ``gpu_id`` unless it is resident on
        another device already (:135-137).  ``remote`` has no effect in the reference either (it is only stored).
        ``gpu_id`` is the CUDA ordinal the reference's dispatch_local computes (expert_id % torch.cuda.device_count(),
        expert_executor.py:49-54); a device this handle does not drive (one process per GPU on a box where the process
        still sees all of them) means the expert's home engine: see slot_for_gpu()."""
        h = self.handle
        key = (int(layer_idx), int(expert_idx))
        if key not in h._expert_ids:
            raise RuntimeError(f"ExpertDispatcher::Enqueue: expert (layer {key[0]}, expert {key[1]}) was never registered")
        home = h._slot_of.get(key)
        slot = slot_for_gpu(h.devices, int(gpu_id), home)
        if home is not None and home != slot and h.engines[home] is not None and h.engines[home].is_resident(*key):
            slot = home  # "if (expert_node->node->device.is_cuda()) args.gpu_id = device.index()"
        if slot != home:
            self._place(key[0], key[1], slot)  # a device the expert has not been on yet: its arena gets the blob now
            h._slot_of[key] = slot
        self._queue.append((key[0], key[1], slot))

    def wait_expert(self) -> List[Tuple[torch.Tensor, int, int, int]]:
        """WaitExpert (expert_dispatcher.cpp:436-450): [(output [t_e, H], layer, expert, hit)] in ascending expert id, every
        output on the hidden states' device (OutputFunc, :397-434)."""
        queue, self._queue = self._queue, []
        if len(queue) != self._expected:
            raise RuntimeError(f"expected {self._expected} enqueued experts, got {len(queue)}")
        if not queue:
            return []
        layers = {l for l, _, _ in queue}
        if len(layers) != 1:
            raise RuntimeError("one wait_expert() serves one layer (as dispatch_local uses it)")
        layer = layers.pop()
        h = self.handle
        if not h._live_engines():
            raise RuntimeError("no expert was registered")
        home_dev = self._hidden.device
        by_slot: Dict[int, List[int]] = {}
        for _, e, slot in queue:
            by_slot.setdefault(slot, []).append(e)

        def run(slot: int):
            eng = h.engines[slot]
            dev = torch.device("cuda", h.devices[slot])
            with torch.cuda.device(dev):
                hidden = self._hidden if self._hidden.device == dev else self._hidden.to(dev)   # GPUFetchFunc: input.to(node device)
                mask = self._mask if self._mask.device == dev else self._mask.to(dev)
                enq = sorted(set(by_slot[slot]))
                if hidden.shape[0] > eng.cfg.max_tokens:
                    eng.reserve_tokens(hidden.shape[0])
                # only the experts enqueued on this device run here: the id list goes down with the call (no second mask, no
                # host-to-device copy on this side of the boundary)
                y, counts, hit = eng.dispatch_mask(layer, hidden, mask, experts=None if len(enq) == mask.shape[1] else enq)
                if dev != home_dev:
                    y = y.to(home_dev)  # OutputFunc: output.to(output_device)
                    torch.cuda.current_stream(dev).synchronize()
            return y, counts, hit

        slots = sorted(by_slot)
        if len(slots) == 1:
            results = [run(slots[0])]
        else:  # the devices work concurrently (the reference: one fetch + exec thread set per GPU); ctypes drops the GIL in the call
            with ThreadPoolExecutor(max_workers=len(slots)) as pool:
                results = list(pool.map(run, slots))
        out = []
        for y, counts, hit in results:
            row = 0
            for e in range(len(counts)):
                if counts[e] > 0:
                    out.append((y[row:row + counts[e]], layer, e, int(hit[e])))
                    row += int(counts[e])
        out.sort(key=lambda r: r[2])
        return out

    def clear_expert_cache_counts(self):
        for eng in self.handle._live_engines():
            eng.clear_expert_cache_counts()
