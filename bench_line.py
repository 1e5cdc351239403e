"""The ONE stdout line of bench.py, kept small enough for the driver to parse.

Round 5's line grew to 24 KB (prose strings repeated in every parity object and offload sub-leg) and the driver's
parser gave up: BENCH_r05.parsed = null.  This module owns the contract now:

  * `compact(full)`   the <= 8 KB line: the contract's keys, `roofline`, `cpu_baseline`, and a few numbers per leg;
  * `write_details()` everything bench.py measured (the old 24 KB object) into `bench_details.json` beside bench.py and,
                      when it exists, `gpurun_out/bench_details.json` so it travels back from the GPU box;
  * `check(line)`     the invariants tests/test_bench_line_cpu.py and tests/test_gpu_bench_ranks.py assert.

Pure Python, no torch: the CPU suite exercises it on canned result objects (profiles/r05_bench_default_mixtral8x7b.json
is the 24 KB object that broke the parser).
"""
import json
import os

MAX_LINE_BYTES = 8192

REQUIRED = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
            "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline")

_ROOF_KEYS = ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "traffic_over_algorithmic", "traffic_ok",
              "avg_launch_us", "bytes_per_launch", "timed_by", "empty_event_interval_us", "frac_minus_empty_event_interval")
_CPU_KEYS = ("value", "unit", "cores", "kind", "ms_per_token", "host_cores", "sample")
_REFC_KEYS = ("value", "unit", "cores", "kind", "ms_per_token")
_PARITY_KEYS = ("ok", "routing_bit_exact", "mean_rel_err", "max_rel_err", "worst_err_over_bar", "pairs_checked", "path")
_MISS_KEYS = ("ms_per_token", "hit_rate", "misses_per_token", "h2d_GBps", "h2d_frac_of_pcie5_x16", "overlap",
              "prefetch_issued", "prefetch_useful", "prefetch_precision", "speculation_kind")
_KERNEL_KEYS = ("avg_launch_us", "bytes_per_launch", "frac_of_hbm_peak")


def _pick(src, keys):
    if not isinstance(src, dict):
        return None
    return {k: src[k] for k in keys if k in src and src[k] is not None}


def _short(s, n):
    if not isinstance(s, str) or len(s) <= n:
        return s
    return s[: n - 1] + "…"


def _roofline(r):
    out = _pick(r, _ROOF_KEYS)
    if out is None:
        return None
    out.setdefault("traffic", None)  # the contract: HBM bytes from the PMC counters, or null
    if "kernel" in out:
        out["kernel"] = _short(out["kernel"].split(":")[0], 64)
    return out


def _cpu(c):
    out = _pick(c, _CPU_KEYS)
    if out is None:
        return None
    if "sample" in out:
        out["sample"] = _short(out["sample"], 160)
    rc = c.get("reference_compiled") if isinstance(c, dict) else None
    if isinstance(rc, dict):
        o = _pick(rc, _REFC_KEYS)
        o["bit_identical"] = rc.get("bit_identical_to_the_restatement", rc.get("bit_identical"))
        out["reference_compiled"] = o
    return out


def _parity(p):
    out = _pick(p, _PARITY_KEYS)
    if out is None:
        return None
    arm = p.get("fp32_exact_arm")
    if isinstance(arm, dict):
        out["exact_arm_ratio"] = arm.get("ratio_over_the_sample")
        out["exact_arm_ok"] = arm.get("ok")
    return out


def _miss(m):
    out = _pick(m, _MISS_KEYS)
    if out is None:
        return None
    if "ms_per_token_over_pcie_bound" in m:
        out["over_pcie_bound"] = m["ms_per_token_over_pcie_bound"]
    subs = m.get("sub_legs")
    if isinstance(subs, list) and subs:
        out["sub_legs"] = len(subs)
        # speculation on the residual stream (where the next layer is predictable): on demand | EAM history | next-layer gate
        res = [s for s in subs if isinstance(s, dict) and str(s.get("activations", "")).startswith("residual")]
        if res:
            out["residual_stream"] = [{"routing": s.get("routing"), "kind": s.get("speculation_kind"), "ms_per_token": s.get("ms_per_token"),
                                       "hit_rate": s.get("hit_rate"), "overlap": s.get("overlap"), "precision": s.get("prefetch_precision")} for s in res]
        else:
            best = [s for s in subs if isinstance(s, dict) and s.get("prefetch_issued")]
            if best:  # the speculating sub-leg, beside the on-demand headline of this object
                b = min(best, key=lambda s: s.get("ms_per_token", 1e30))
                out["speculating"] = _pick(b, ("routing", "policy", "ms_per_token", "hit_rate", "overlap", "prefetch_issued",
                                               "prefetch_useful", "h2d_GBps", "speculation_kind"))
    return out


def _workload_short(w):
    """'DeepSeek-V2-Lite MoE layers: L=26 ...' -> 'DeepSeek-V2-Lite'"""
    if not isinstance(w, str):
        return w
    return w.split(" MoE layers")[0].split(",")[0].strip()


def _other(o):
    if not isinstance(o, dict):
        return None
    if "error" in o:
        return {"workload_short": _workload_short(o.get("workload")), "error": _short(o["error"], 120)}
    pr = o.get("parity") or {}
    out = {"workload_short": _workload_short(o.get("workload")), "dtype": o.get("dtype", o.get("dtype_short")),
           "batch": o.get("batch"), "ms_per_step": o.get("ms_per_step"),
           "frac_whole_step": o.get("frac_of_hbm_peak_whole_step"), "parity_ok": pr.get("ok")}
    if o.get("mean_rel_err") is not None:  # the fp16 legs: north_star's literal tolerance beside the measured numbers
        out["mean_rel_err"], out["max_rel_err"] = o.get("mean_rel_err"), o.get("max_rel_err")
    for stage in ("ffn_stage1", "ffn_stage2"):
        k = o.get(stage)
        if isinstance(k, dict) and k.get("frac_of_hbm_peak") is not None:
            out[stage + "_frac"] = k["frac_of_hbm_peak"]
    off = o.get("offload_regime")
    if isinstance(off, dict):
        out["offload"] = _miss(off)
    return {k: v for k, v in out.items() if v is not None}


def compact(full):
    """the driver's line from everything bench.py measured (`full`, the object that goes to bench_details.json)"""
    line = {k: full.get(k) for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better",
                                     "scaling", "vs_baseline", "dtype", "data")}
    cfg = full.get("config") or {}
    line["config"] = {k: cfg[k] for k in ("workload", "parallelism", "cache_policy", "per_token_decode_latency_ms") if k in cfg}
    line["roofline"] = _roofline(full.get("roofline"))
    line["cpu_baseline"] = _cpu(full.get("cpu_baseline"))
    line["parity"] = _parity(full.get("parity"))
    if full.get("windows_ms"):
        line["windows_ms"] = full["windows_ms"]
    ks = full.get("kernels")
    if isinstance(ks, dict):
        line["kernels"] = {name: _pick(k, _KERNEL_KEYS) for name, k in ks.items() if isinstance(k, dict)}
    pf = full.get("prefill")
    if isinstance(pf, dict):
        line["prefill"] = _pick(pf, ("tokens", "ms_all_layers", "tokens_per_s"))
        pk = pf.get("kernels") or {}
        for stage in ("ffn_stage1", "ffn_stage2"):
            if isinstance(pk.get(stage), dict):
                line["prefill"][stage + "_frac"] = pk[stage].get("frac_of_hbm_peak")
    ps = full.get("prefetch_stream")
    if isinstance(ps, dict):
        line["prefetch_stream"] = _pick(ps, ("GiB", "GBps", "frac_of_pcie5_x16_63GBps", "frac_of_hbm_peak"))
    if isinstance(full.get("miss_heavy"), dict):
        line["miss_heavy"] = _miss(full["miss_heavy"])
    if isinstance(full.get("dropin"), dict):  # {workload: tools/dropin_time.measure()}
        line["dropin"] = {_workload_short(k): (_pick(v, ("ms_per_token", "fused_ms_per_token", "over_fused", "host_us_per_call", "calls_per_token",
                                                          "boundary_us_per_layer", "reference_python_us_per_layer", "parity_ok"))
                                               if "error" not in v else {"error": _short(v["error"], 100)})
                          for k, v in full["dropin"].items() if isinstance(v, dict)}
    if isinstance(full.get("prefill_4096"), dict):
        line["prefill_4096"] = _pick(full["prefill_4096"], ("tokens", "ms_all_layers", "gated_PFLOPs", "down_PFLOPs", "frac_of_mfma_peak"))
    ep = full.get("ep_transport")
    if ep is not None:
        line["ep_transport"] = _pick(ep, ("chosen", "ms_per_step_by_transport", "world")) if isinstance(ep, dict) else _short(str(ep), 200)
    if isinstance(full.get("ep_phases_us_per_layer"), dict):
        line["ep_phases_us_per_layer"] = {k: v for k, v in full["ep_phases_us_per_layer"].items() if isinstance(v, (int, float))}
    others = full.get("other_configs")
    if others:
        line["other_configs"] = [x for x in (_other(o) for o in others) if x]
    line["details"] = full.get("details_file", "bench_details.json")
    # last resort: a line that still does not fit sheds its optional parts, least important first
    for drop in ("kernels", "ep_phases_us_per_layer", "prefetch_stream", "windows_ms", "prefill", "prefill_4096", "dropin", "miss_heavy", "other_configs"):
        if len(json.dumps(line)) < MAX_LINE_BYTES:
            break
        line.pop(drop, None)
    return line


def check(line):
    """raises AssertionError unless `line` (dict or its JSON text) honours the driver's contract"""
    text = line if isinstance(line, str) else json.dumps(line)
    obj = json.loads(text)
    assert "\n" not in text.strip(), "one line"
    assert len(text.encode()) < MAX_LINE_BYTES, f"bench line is {len(text.encode())} B >= {MAX_LINE_BYTES}"
    for k in REQUIRED:
        assert k in obj, f"bench line lacks {k!r}"
    assert isinstance(obj["config"], dict) and "workload" in obj["config"] and "model" not in obj["config"]
    roof = obj["roofline"]
    if roof is not None:
        for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
            assert k in roof, f"roofline lacks {k!r}"
        assert abs(roof["frac"] - roof["achieved"] / roof["peak"]) < 2e-3
    cpu = obj["cpu_baseline"]
    if cpu is not None:
        for k in ("value", "unit", "cores", "kind", "sample"):
            assert k in cpu, f"cpu_baseline lacks {k!r}"
    return obj


def write_details(full, here):
    """the full object beside bench.py (and under gpurun_out/ when that directory exists, so it comes back from the box)"""
    paths = [os.path.join(here, "bench_details.json")]
    scratch = os.path.join(here, "gpurun_out")
    if os.path.isdir(scratch):
        paths.append(os.path.join(scratch, "bench_details.json"))
    written = []
    for p in paths:
        try:
            with open(p, "w") as f:
                json.dump(full, f, indent=1)
            written.append(p)
        except OSError:
            pass
    return written
