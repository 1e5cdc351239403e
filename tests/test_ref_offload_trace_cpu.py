"""tests/golden/offload_engine_trace.json must be what the REFERENCE'S code does today: re-run the recording (the
reference's OffloadEngine._offload_state_dict / setup_archer_hooks / module hooks, its SyncMixtralSparseMoeBlock and
dispatch_local, executed from /root/reference by oracle/gen_offload_trace.py against the recording stand-in of the
pybind module) and compare it with the committed fixture.  Needs /root/reference (this container); the GPU test
tests/test_gpu_dropin.py::test_prefetch_op_under_the_reference_offload_engines_own_call_sequence then holds the real
prefetch_op to this sequence."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden", "offload_engine_trace.json")


def test_fixture_is_well_formed():
    d = json.load(open(GOLD))
    names = [c[0] for c in d["calls"]]
    assert names.count("forward") == 2 and names.count("set_topology") == 1
    L, E = d["shapes"]["L"], d["shapes"]["E"]
    assert names.count("register_expert") == L * E and names.count("offload") == names.count("register") == len(d["name_id_map"])
    assert names.count("begin") == names.count("end") and names.count("wait_expert") == 2 * L
    assert [n for n, _ in d["topology"]] == [x for l in range(L) for x in (f"layers.{l}", f"layers.{l}.block_sparse_moe.experts")] + ["lm_head"]


@pytest.mark.skipif(not os.path.isdir("/root/reference"), reason="needs /root/reference")
def test_fixture_matches_a_fresh_run_of_the_reference_code():
    code = ("import json,sys; sys.path.insert(0, %r); from oracle import gen_offload_trace as g; "
            "print('TRACE' + json.dumps(g.record()))" % ROOT)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300,
                       env=dict(os.environ, PYTHONDONTWRITEBYTECODE="1"), cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    fresh = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("TRACE")][-1][5:])
    gold = json.load(open(GOLD))
    assert fresh["calls"] == gold["calls"]
    assert fresh["topology"] == gold["topology"] and fresh["name_id_map"] == gold["name_id_map"]
    assert fresh["output"] == gold["output"]
