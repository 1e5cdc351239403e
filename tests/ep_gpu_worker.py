"""Worker of tests/test_gpu_ep_processes.py: one expert-parallel rank (its own process, its own HIP engine) on GPU 0.
Launched by torch.distributed.run with the gloo backend (RCCL refuses two ranks on one GPU).  EP_TRANSPORT=peer-store
(default): the product's direct exchange — every rank maps the other processes' windows with hipIpc* and stores its rows
straight into them; the process group only carries the 192-byte bootstrap blobs and the all-reduced verdicts.
EP_TRANSPORT=torch: all_to_all_single with the rows staged through host memory (the older form of this test)."""
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo", rank=rank, world_size=world)
    transport = os.environ.get("EP_TRANSPORT", "peer-store")
    from helpers import R, acts, assert_block_close, make_weights
    from moe_infinity_amd import MoEEngine
    from moe_infinity_amd import config as Cf
    from moe_infinity_amd.ep import ExpertParallelMoE, HipEpOps

    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    # EP_POLL_ONLY_RANK=r: MOEINF_EP_PEER_POLL=1 in the environment of rank r ALONE.  The exchange form is a group decision
    # (every rank derives it from all ranks' blobs, csrc/ep_peer.h): rank r's wish is out-voted by the ranks that share the GPU
    # with it and did not ask, so ALL ranks run wait kernels + the routed form — not rank r the broadcast form on its own.
    only = os.environ.get("EP_POLL_ONLY_RANK", "")
    if only != "":
        os.environ.pop("MOEINF_EP_PEER_POLL", None)
        if int(only) == rank:
            os.environ["MOEINF_EP_PEER_POLL"] = "1"
    for family, e, k, n_shared in (("mixtral", 8, 2, 0), ("deepseek", 16, 4, 2)):
        h, f, L, tmax = 256, 512, 2, 40
        ws = [make_weights(family, h, f, e, 3100 + 10 * l, torch.bfloat16, n_shared=n_shared) for l in range(L)]
        et, rk = (Cf.EXPERT_MIXTRAL, Cf.ROUTER_MIXTRAL) if family == "mixtral" else (Cf.EXPERT_DEEPSEEK, Cf.ROUTER_DEEPSEEK)
        eng = MoEEngine(Cf.EngineConfig(num_layers=L, num_experts=e, expert_type=et, hidden=h, inter=f, top_k=k, router_kind=rk,
                                        dtype=Cf.DTYPE_BF16, shared_inter=f * n_shared, device_memory_ratio=0.25, max_tokens=tmax * world,
                                        ep_rank=rank, ep_size=world))
        for l in range(L):
            for i, ex in enumerate(ws[l][1]):
                if i % world == rank:
                    eng.register_expert(l, i, ex)
            if ws[l][2]:
                eng.register_shared(l, ws[l][2])
        # uniform_tokens: one-token forwards (the same count on every rank below) take the broadcast form when the consumer
        # kernels poll for themselves (MOEINF_EP_PEER_POLL=1); with wait kernels the routed form runs
        ep = ExpertParallelMoE(HipEpOps(eng), h, k, tmax, torch.bfloat16, dev, var_threshold=64, num_experts=e, transport=transport,
                               uniform_tokens=os.environ.get("EP_UNIFORM", "0") == "1")
        assert ep.transport == transport, f"rank {rank}: wanted {transport}, got {ep.transport}: {ep.native_note}"
        if transport == "peer-store":
            t = eng.ep_transport()  # the ranks share GPU 0: detected from the PCI bus ids in the blobs
            want_poll = os.environ.get("MOEINF_EP_PEER_POLL", "0") == "1" and only == ""  # default for ranks that share a GPU: wait kernels
            assert t["transport"] == "peer-store" and t["shared_device"] and t["poll_in_kernels"] == want_poll, t
        dist.barrier()
        for t in (1, 1, 3 + rank, tmax - 3 * rank, 1):  # batch 1 (twice: decision path, then sync-free), ragged small batches (fixed form), prefill-sized (variable split), batch 1 again
            for l in range(L):
                x = acts(t, h, torch.bfloat16, 3200 + 7 * t + l + 1000 * rank)
                out = ep.forward(l, x.to(dev), ws[l][0].to(dev))
                eng.sync()  # a poll of the exchange that gave up raises here, before its output is looked at
                out = out.cpu()
                if family == "mixtral":
                    ref = R.block_mixtral(x[None], ws[l][0], ws[l][1], top_k=k)
                else:
                    ref = R.block_deepseek(x[None], ws[l][0], ws[l][1], k, shared=ws[l][2])
                assert_block_close(out, ref, torch.bfloat16, f"rank {rank} {family} t={t} layer {l} ({ep.last_form})")
                assert ep.last_form == ("variable" if t * k > 64 else "fixed")
        eng.sync()      # a peer-store poll that timed out would raise here
        dist.barrier()  # nobody frees a window another rank may still store into
        eng.close()
        dist.barrier()
    print(f"EP_WORKER_OK rank {rank} of {world} transport {transport}", flush=True)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
