#!/usr/bin/env python3
"""Timeline of the fused decode launches (csrc/layer_fused.hip: the gate + stage-1 front of the gated families, the one-launch
Switch layer): runs a few batch-1 forwards with MOEINF_LAYER1_TRACE set (the engine writes the per-workgroup timestamps of the LAST launch at destroy) and prints, per role,
when its workgroups started / got past their waits / finished (microseconds after the first workgroup started).
usage: layer1_trace.py [--switch | --mixtral] [out.txt]   (--switch: Switch-base-8, the one-launch form that is its default; extra MOEINF_*
knobs are taken from the environment)"""
import os, statistics, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import sys, torch
sys.path.insert(0, %r)
from moe_infinity_amd import MoEEngine, config as Cf
from oracle.synth import acts
cfg = {"switch": Cf.switch_base_8, "mixtral": Cf.mixtral_8x7b, "deepseek": Cf.deepseek_v2_lite}[sys.argv[1]](device_memory_ratio=0.5, max_tokens=1)
cfg.num_layers = 2
eng = MoEEngine(cfg)
dev = torch.device("cuda:0")
off, siz, tot = eng.expert_layout(0)
for l in range(2):
    for e in range(cfg.num_experts):
        eng.register_expert(l, e, None)
        es = 4 if eng.dtype == torch.float32 else 2
        eng.expert_host_view(l, e).view(eng.dtype).copy_(torch.empty(tot // es, dtype=eng.dtype, device=dev).normal_(0, 0.02))
    if cfg.shared_inter:
        _, sizs, _ = eng.expert_layout(1)
        eng.register_shared(l, [torch.empty(s // 2, dtype=eng.dtype).normal_(0, 0.02) for s in sizs])
    eng.prefetch(l, list(range(cfg.num_experts)))
eng.sync_copies()
gates = [(torch.randn(cfg.num_experts, cfg.hidden, device=dev) * (0.5 if sys.argv[1] == "switch" else 0.02)).to(eng.gate_dtype) for _ in range(2)]
x = acts(1, cfg.hidden, eng.dtype, 11).to(dev)
out = torch.empty(1, cfg.hidden, dtype=eng.dtype, device=dev)
for i in range(40):
    eng.forward(i %% 2, x, gates[i %% 2], out=out)
torch.cuda.synchronize()
eng.sync()
eng.close()
''' % ROOT
def main():
    path = tempfile.mktemp(suffix=".l1trace")
    env = dict(os.environ, MOEINF_LAYER1_TRACE=path)
    model = "deepseek"
    for m in ("switch", "mixtral"):
        if "--" + m in sys.argv:
            model = m
            sys.argv.remove("--" + m)
    r = subprocess.run([sys.executable, "-c", CHILD, model], env=env, capture_output=True, text=True)
    if r.returncode or not os.path.exists(path):
        print("child failed:", r.stderr[-1500:]); return 1
    rows = [[int(v) for v in ln.split()] for ln in open(path)]  # workgroup item role index t0 t1 t2 t3
    names = {1: "gate", 2: "shared stage 1", 3: "meta", 4: "routed stage 1", 5: "shared stage 2", 6: "routed stage 2"}
    t0 = min(r[4] for r in rows if r[4])
    us = lambda t: (t - t0) / 100.0
    out = []
    for role in sorted(names):
        part = [r for r in rows if r[2] == role]
        if not part:
            continue
        def col(k):
            v = [us(r[4 + k]) for r in part if r[4 + k]]
            return "      -      " if not v else f"{min(v):6.1f} {statistics.median(v):6.1f} {max(v):6.1f}"
        out.append(f"{names[role]:16s} n={len(part):4d} | start {col(0)} | wait1 over {col(1)} | wait2 over {col(2)} | end {col(3)}   (min median max, us)")
    nwg = 1 + max(r[0] for r in rows)
    per = [sum(1 for r in rows if r[0] == w) for w in range(nwg)]
    ends = [max(us(r[7]) for r in rows if r[0] == w) for w in range(nwg) if per[w]]
    out.append(f"workgroups {nwg}, items per workgroup {min(per)}..{max(per)}; last item of a workgroup ends {min(ends):.1f} / {statistics.median(ends):.1f} / {max(ends):.1f} us (min / median / max)")
    out.append(f"launch span: {us(max(r[7] for r in rows if r[7])):.1f} us")
    txt = "\n".join(out)
    print(txt)
    if len(sys.argv) > 1:
        open(sys.argv[1], "w").write(txt + "\n")
    os.unlink(path)
    return 0
if __name__ == "__main__":
    sys.exit(main())
