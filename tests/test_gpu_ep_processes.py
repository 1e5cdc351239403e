"""Expert parallelism with REAL processes on the GPU: two (and three) ranks, each its own process with its own HIP
engine holding the experts e % world == rank, all on GPU 0 of a one-GPU box.  The decode-sized exchanges run over the
product's own transport, the direct peer-store exchange (moeinf_ep_peer_*): every process maps the others' windows with
hipIpcOpenMemHandle and its kernels store rows straight into them — the multi-rank path of the product, executed between
real processes (RCCL refuses several ranks per GPU, "Duplicate GPU detected"; gloo only carries the bootstrap blobs and the
prefill-sized, variable-split exchanges, staged through host memory).  One case keeps the older all-host-staged form.
Every rank checks its own tokens against the oracle with the block bar."""
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("world,transport,poll,uniform", [(2, "peer-store", "0", "0"), (3, "peer-store", "0", "0"), (2, "peer-store", "1", "0"),
                                                          (2, "peer-store", "1", "1"), (3, "peer-store", "1", "1"), (2, "torch", "", "0"),
                                                          (3, "peer-store", "only_rank_1", "1")],
                         ids=["world2_peer_store_wait_kernels", "world3_peer_store_wait_kernels", "world2_peer_store_polls_inside_the_consumer_kernels",
                              "world2_batch1_broadcast_form", "world3_batch1_broadcast_form", "world2_torch_host_staged",
                              "world3_one_rank_alone_asks_for_polls_and_is_outvoted"])
def test_expert_parallel_ranks_as_processes_on_one_gpu(world, transport, poll, uniform):
    """poll: MOEINF_EP_PEER_POLL — "0" = the mode ranks that share a GPU get by default (one-wave wait kernels), "1" = the mode
    of one rank per GPU (the consumer kernels poll their flag words themselves), forced here so that it, too, has run between
    real processes (the test shapes leave most of the GPU idle, so a polling kernel cannot starve the rank it waits for)."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "tests", "ep_gpu_worker.py")]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", EP_TRANSPORT=transport)
    if poll.startswith("only_rank_"):  # the knob in ONE rank's environment: the group must still agree on one exchange form
        env["EP_POLL_ONLY_RANK"] = poll[len("only_rank_"):]
    elif poll:
        env["MOEINF_EP_PEER_POLL"] = poll
    env["EP_UNIFORM"] = uniform  # "1": every rank promises equal token counts -> one-token forwards take the broadcast form
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    # the ranks share one stdout: their lines can run together
    assert r.returncode == 0 and r.stdout.count("EP_WORKER_OK") == world and r.stdout.count(f"transport {transport}") == world, (r.stdout[-2000:] + "\n" + r.stderr[-4000:])
