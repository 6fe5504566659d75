"""Assembles DESIGN.md from its parts (docs/design_parts/*.md) and fills the measured-numbers placeholders of section 8 / 9 / 11 from a
bench.py JSON line:  python scripts/make_design.py profiles/r06_bench_cfg2.json docs/design_parts/graph_destroy.txt"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = os.path.join(ROOT, "docs", "design_parts")
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
gd = open(sys.argv[2]).read().strip() if len(sys.argv) > 2 else "(not run)"
al = d["also"]
r = d["roofline"]


def conv(x):
    rr = x["roofline"]
    ex = rr.get("executed", {}).get("frac", rr.get("executed_frac"))
    return f"{rr['achieved']:.0f} TFLOP/s = **{rr['frac']:.3f}** of 2.5 PF algorithmic, {ex:.3f} executed, {rr['ms_per_step']:.1f} ms"


def score(x):
    s = x.get("score_gemm")
    return f"{s['us_per_step']:.0f} µs = {s['achieved']:.0f} TFLOP/s = **{s['frac']:.3f}**" if s else ""


cb = d.get("cpu_baseline", {})
rep = {
    "CFG2_VALUE": f"{d['value']:.0f}", "CFG2_MS": f"{d['ms_per_step']:.2f}", "CFG2_CONV": conv(d),
    "CFG2_HBM": f"{d['hbm_family']['achieved'] / 1e3:.2f} TB/s = **{d['hbm_family']['frac']:.2f}** of 8 TB/s, {d['hbm_family']['ms_per_step']:.2f} ms ({d['hbm_family']['launches_per_step']} launches)",
    "CFG2_SCORE": score(d), "CFG2_SCORE_FRAC": f"{d['score_gemm']['frac']:.3f}",
    "ONE_VALUE": f"{al['one_stream']['value']:.0f}", "ONE_MS": f"{al['one_stream']['ms_per_step']:.2f}",
    "TWO_MS": f"{al['one_stream']['two_stream_ms_per_step']:.2f}", "TWO_SPEEDUP": f"{al['one_stream']['two_stream_speedup']:.3f}",
    "MOD_VALUE": f"{al['module']['value']:.0f}", "MOD_MS": f"{al['module']['ms_per_step']:.2f}", "MOD_FRAC": f"**{al['module']['vs_engine_path']:.3f}**",
    "F32_VALUE": f"{al['f32']['value']:.0f}", "F32_MS": f"{al['f32']['ms_per_step']:.1f}",
    "X6_VALUE": f"{al['f32_bf16x6']['value']:.0f}", "X6_MS": f"{al['f32_bf16x6']['ms_per_step']:.1f}",
    "CFG4_VALUE": f"{al['cfg4']['value']:.0f}", "CFG4_MS": f"{al['cfg4']['ms_per_step']:.1f}", "CFG4_CONV": conv(al["cfg4"]), "CFG4_SCORE": score(al["cfg4"]),
    "CFG4_SCORE_FRAC": f"{al['cfg4']['score_gemm']['frac']:.3f}",
    "CFG5_VALUE": f"{al['cfg5']['value']:.0f}", "CFG5_MS": f"{al['cfg5']['ms_per_step']:.1f}", "CFG5_CONV": conv(al["cfg5"]), "CFG5_SCORE": score(al["cfg5"]),
    "CFG5_SCORE_FRAC": f"{al['cfg5']['score_gemm']['frac']:.3f}",
    "CFG5F_VALUE": f"{al['cfg5_fused_score']['value']:.0f}", "CFG5F_MS": f"{al['cfg5_fused_score']['ms_per_step']:.1f}", "CFG5F_SCORE": score(al["cfg5_fused_score"]),
    "CPU": f"{cb.get('value')} clips/s at {cb.get('cores')} threads ({cb.get('sample', '')[:120]}…)",
    "TRAFFIC": f"{r['traffic']}", "ALG_GB": f"{r['algorithmic_GB_per_launch']}", "TRAFFIC_RATIO": f"{r['traffic'] / r['algorithmic_GB_per_launch']:.2f}" if r.get("traffic") else "n/a",
    "GRAPH_DESTROY": gd,
    "SECTION10": open(os.path.join(P, "10_beyond.md")).read().split("\n", 2)[2].strip(),
}
out = ""
for part in ("00_head.md", "04_kernels.md", "05_to_07.md", "08_on.md"):
    out += open(os.path.join(P, part)).read().rstrip() + "\n\n"
for k, v in rep.items():
    out = out.replace(f"@{k}@", v)
left = [w for w in out.split() if w.startswith("@") and w.endswith("@") and len(w) > 2]
assert not left, left
open(os.path.join(ROOT, "DESIGN.md"), "w").write(out.rstrip() + "\n")
print(len(out), "bytes")
