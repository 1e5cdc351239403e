"""The drop-in boundary on CPU: our mirror of ``DistributedExpertExecutor.dispatch_local`` (12 lines restated from
moe_infinity/distributed/expert_executor.py:32-58 — an interface mirror, not a design of ours) must drive a dispatcher
exactly like the reference's own function does.  Runs the REFERENCE'S code from /root/reference when it is there
(build container); the GPU tests then use the mirror with the real dispatcher."""
import os

import pytest
import torch

REF = "/root/reference"


class Recorder:
    """expert_dispatcher stand-in that records the call sequence"""

    def __init__(self):
        self.calls = []

    def set_inputs(self, hidden, mask):
        self.calls.append(("set_inputs", tuple(hidden.shape), tuple(mask.shape), int(mask.sum())))

    def set_expected_queue(self, n):
        self.calls.append(("set_expected_queue", int(n)))

    def enqueue_expert(self, layer, expert, gpu, remote):
        self.calls.append(("enqueue_expert", int(layer), int(expert), int(gpu), bool(remote)))

    def wait_expert(self):
        self.calls.append(("wait_expert",))
        return [("result", len(self.calls))]


@pytest.mark.skipif(not os.path.isdir(REF), reason="needs /root/reference")
def test_dispatch_local_mirror_drives_the_dispatcher_like_the_reference_does(monkeypatch):
    from moe_infinity_amd.expert_executor import DistributedExpertExecutor as Ours
    from oracle.gen_golden import import_reference

    mods = import_reference()
    Theirs = mods["executor"].DistributedExpertExecutor
    monkeypatch.setattr(torch.cuda, "device_count", lambda: 3)  # gpu_id = expert_id % device_count
    g = torch.Generator().manual_seed(3)
    for trial in range(25):
        t, e = int(torch.randint(1, 40, (1,), generator=g)), int(torch.randint(2, 130, (1,), generator=g))
        mask = torch.rand(t, e, generator=g) < (2.0 / e)
        if trial % 5 == 0:
            mask[:] = False  # nothing routed: empty expert list
        hidden = torch.randn(t, 16, generator=g)
        a, b = Recorder(), Recorder()
        ra, rb = Theirs(None), Ours(None)
        ra.set_expert_dispatcher(a)
        rb.set_expert_dispatcher(b)
        layer = trial % 7
        out_a = ra.dispatch_local(hidden, mask, layer)
        out_b = rb.dispatch_local(hidden, mask, layer)
        assert a.calls == b.calls and out_a == out_b
        # 3-D masks ([B, S, E]) are viewed (-1, E) by both
        out_a = ra.dispatch_local(hidden[None], mask[None], layer)
        out_b = rb.dispatch_local(hidden[None], mask[None], layer)
        assert a.calls == b.calls and out_a == out_b


def test_prefetch_op_exports_the_pybind_surface():
    """every method name of py_archer_prefetch.cpp:14-92 exists on the drop-in classes with the reference's arity"""
    import inspect

    from moe_infinity_amd import prefetch_op as P

    want_handle = {"offload": 2, "register": None, "set_tensor_device": 2, "begin": 2, "end": 2, "get_hit_rate": 0, "set_trace": 1,
                   "set_topology": 1, "update_tensor_map": 2, "is_tensor_offloaded": 1, "is_tensor_index_initialized": 0,
                   "is_tensor_on_device": 1, "get_node_default_device": 1, "get_node_device": 1, "prefetch_tensors": 2,
                   "replace_cache_candidates": 1, "enqueue_prefetch": 2, "fetch_tensors": 2, "clean_up_resources": 0}
    for name, arity in want_handle.items():
        fn = getattr(P.prefetch_handle, name)
        if arity is not None:
            assert len([p for p in inspect.signature(fn).parameters.values() if p.name != "self"]) == arity, name
    assert list(inspect.signature(P.prefetch_handle.__init__).parameters)[1:] == ["prefix", "device_memory_ratio"]
    assert list(inspect.signature(P.expert_dispatcher.__init__).parameters)[1:] == ["num_experts", "num_layers", "dtype", "expert_type", "num_threads"]
    want_disp = {"register_expert": 3, "enqueue_expert": 4, "set_inputs": 2, "set_expected_queue": 1, "wait_expert": 0, "clear_expert_cache_counts": 0}
    for name, arity in want_disp.items():
        assert len([p for p in inspect.signature(getattr(P.expert_dispatcher, name)).parameters.values() if p.name != "self"]) == arity, name
    if not torch.cuda.is_available():
        with pytest.raises(RuntimeError, match="no CPU fallback"):
            P.prefetch_handle("/tmp/x", 0.5)
