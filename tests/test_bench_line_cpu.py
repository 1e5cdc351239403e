"""The driver parses ONE stdout line of bench.py.  Round 5's line was 24 KB and BENCH_r05.parsed came back null; this
guards the line bench_line.compact() builds: < 8 KB, every key of the contract, `roofline` and `cpu_baseline` intact.
The canned input is the very object that broke the parser (profiles/r05_bench_default_mixtral8x7b.json)."""
import copy
import json
import os

import pytest

import bench_line

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def full():
    with open(os.path.join(ROOT, "profiles", "r05_bench_default_mixtral8x7b.json")) as f:
        return json.load(f)


def test_the_object_that_broke_the_driver_now_fits(full):
    assert len(json.dumps(full)) > 20000  # the canned object IS the oversized one
    line = bench_line.compact(full)
    text = json.dumps(line)
    assert len(text.encode()) < bench_line.MAX_LINE_BYTES
    obj = bench_line.check(text)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "dtype", "data"):
        assert obj[k] == full[k]
    assert obj["roofline"]["frac"] == full["roofline"]["frac"] and obj["roofline"]["traffic"] == full["roofline"]["traffic"]
    assert obj["cpu_baseline"]["kind"] == "port" and obj["cpu_baseline"]["reference_compiled"]["bit_identical"] is True
    assert obj["parity"]["ok"] and obj["parity"]["routing_bit_exact"]
    assert obj["miss_heavy"]["over_pcie_bound"] == full["miss_heavy"]["ms_per_token_over_pcie_bound"]
    assert [o["workload_short"] for o in obj["other_configs"]] == ["DeepSeek-V2-Lite", "NLLB-MoE-54B", "Switch-base-8", "Mixtral-8x7B",
                                                                  "DeepSeek-V2-Lite", "NLLB-MoE-54B"]
    assert all(o["parity_ok"] for o in obj["other_configs"])


def test_a_line_with_ten_times_the_legs_still_fits(full):
    big = copy.deepcopy(full)
    big["other_configs"] = big["other_configs"] * 10
    big["kernels"] = {f"k{i}": dict(v) for i in range(20) for v in [big["kernels"]["ffn_stage1"]]}
    line = bench_line.compact(big)
    bench_line.check(line)  # optional parts are shed, the contract's keys never
    assert line["roofline"] and line["cpu_baseline"]


def test_a_leg_without_traffic_or_baseline_still_passes(full):
    small = copy.deepcopy(full)
    small["roofline"].pop("traffic")
    small["cpu_baseline"] = None
    small["other_configs"] = None
    line = bench_line.compact(small)
    obj = bench_line.check(line)
    assert obj["roofline"]["traffic"] is None and obj["cpu_baseline"] is None


def test_check_refuses_an_oversized_or_incomplete_line(full):
    line = bench_line.compact(full)
    with pytest.raises(AssertionError):
        bench_line.check(dict(line, junk="x" * 9000))
    bad = dict(line)
    bad.pop("roofline")
    with pytest.raises(AssertionError):
        bench_line.check(bad)


def test_bench_py_prints_the_compact_line_only():
    src = open(os.path.join(ROOT, "bench.py")).read()
    assert "bench_line.compact(" in src and "bench_line.check(" in src
    assert src.count("os.write(real_stdout") == 1  # one writer of the driver's stdout


def test_the_offload_leg_gets_every_local_it_needs_from_run_workload():
    """bench_legs/offload.py takes run_workload's state as a namespace built from locals(): every name it lists must be a
    local of that function (a rename on either side would only fail on the GPU box, minutes into a run)."""
    import ast

    from bench_legs import offload

    tree = ast.parse(open(os.path.join(ROOT, "bench.py")).read())
    fn = next(n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name == "run_workload")
    local = {a.arg for a in fn.args.args}
    for n in ast.walk(fn):
        if isinstance(n, ast.Name) and isinstance(n.ctx, ast.Store):
            local.add(n.id)
        elif isinstance(n, ast.alias):
            local.add((n.asname or n.name).split(".")[0])
    missing = [k for k in offload.NEEDS if k not in local]
    assert not missing, missing


def test_every_tool_and_bench_leg_compiles():
    """tools/*.py and bench_legs/*.py only run on the GPU box: a syntax error there costs a GPU lease"""
    import glob

    files = glob.glob(os.path.join(ROOT, "tools", "*.py")) + glob.glob(os.path.join(ROOT, "bench_legs", "*.py")) + [os.path.join(ROOT, "bench.py")]
    assert len(files) > 10
    for f in files:
        compile(open(f).read(), f, "exec")
